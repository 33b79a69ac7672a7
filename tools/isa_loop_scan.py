import re, sys, subprocess, os, glob, collections, shutil, tempfile
lib=os.path.abspath(sys.argv[1]); sym=sys.argv[2]
tmp=tempfile.mkdtemp(); shutil.copy(lib, tmp+'/lib.so')
subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump','--offloading','lib.so'],cwd=tmp,capture_output=True)
for co in glob.glob(tmp+'/lib.so.*gfx950'):
    out=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump','-d','--no-show-raw-insn',co],capture_output=True,text=True).stdout
    lines=out.split('\n')
    st=[i for i,l in enumerate(lines) if l.endswith('<'+sym+'>:')]
    if not st: continue
    start=st[0]; end=start+1
    while end<len(lines) and not re.match(r'^[0-9a-f]+ <', lines[end]): end+=1
    insts=[]; base=None
    for l in lines[start+1:end]:
        m=re.match(r'\s*(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):', l)
        if not m: continue
        a=int(m.group(3),16)
        if base is None: base=a
        t=re.search(r'\+0x([0-9a-f]+)>', l)
        insts.append((a-base,m.group(1),m.group(2),int(t.group(1),16) if t else None))
    # largest backward branch = step loop
    loops=[(t,a) for a,op,ar,t in insts if t is not None and t<a and (op.startswith('s_cbranch') or op=='s_branch')]
    t0=min(t for t,a in loops if a-t>3000); a1=max(a for t,a in loops if t==t0)
    seg=[x for x in insts if t0<=x[0]<=a1]
    c=collections.Counter(x[1] for x in seg)
    print(sym[:60],'loop',hex(t0),hex(a1),'insts',len(seg),'scratch',sum(v for k,v in c.items() if 'scratch' in k),'ds',sum(v for k,v in c.items() if k.startswith('ds_')),'valu',sum(v for k,v in c.items() if k.startswith('v_')))
    # longest straight-line run without branch containing most valu = hot block
    runs=[]; cur=[]
    for x in seg:
        if x[1].startswith('s_cbranch') or x[1]=='s_branch' or x[1]=='s_setpc_b64':
            runs.append(cur); cur=[]
        else: cur.append(x)
    runs.append(cur)
    runs.sort(key=len, reverse=True)
    for r in runs[:3]:
        cc=collections.Counter(x[1] for x in r)
        print('  block',hex(r[0][0]),hex(r[-1][0]),'insts',len(r),'scratch',[ (hex(x[0]),x[1],x[2][:28]) for x in r if 'scratch' in x[1]][:20],'ds',sum(v for k,v in cc.items() if k.startswith('ds_')),'waitcnt',cc.get('s_waitcnt',0))
shutil.rmtree(tmp)
