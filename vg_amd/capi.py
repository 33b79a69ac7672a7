"""ctypes binding of the C ABI in include/vgk.h.

Plumbing only: problems are described with numpy arrays (so a million of them
can be assembled without a Python loop) and handed to the shared library as the
plain-pointer structs the header declares.  The default library is the HIP
product (vg_amd/libvgamd.so); loading fails loudly if it has not been built.
"""
import ctypes
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libvgamd.so")

VGK_OK = 0
VGK_EINVAL, VGK_ENODEV, VGK_ENOMEM, VGK_ETOOLONG, VGK_EOVERFLOW, VGK_EOPS = -1, -2, -3, -4, -5, -6      # include/vgk.h
VGK_GSSW_LOCAL = 0
VGK_GSSW_PINNED = 1
VGK_XDROP_PINNED = 2
VGK_GSSW_TRACEBACK = 16
OP_M, OP_I, OP_D, OP_S = 0, 1, 2, 3
OP_CHARS = "MIDS"

# struct layouts (must match include/vgk.h)
GRAPH_DT = np.dtype([("n_nodes", "<u4"), ("_pad", "<u4"), ("node_len", "<u8"), ("seq", "<u8"),
                     ("pred_off", "<u8"), ("pred_idx", "<u8")])
PROBLEM_DT = np.dtype([("read", "<u8"), ("read_len", "<u4"), ("flags", "<u4"), ("graph", GRAPH_DT), ("pinning", "<u8"),
                       ("max_gap_length", "<u4"), ("reserved", "<u4"), ("qual", "<u8")])
RESULT_DT = np.dtype([("score", "<i4"), ("status", "<i4"), ("end_node", "<i4"), ("end_offset", "<i4"),
                      ("end_read", "<i4"), ("first_offset", "<i4"), ("n_ops", "<u4"), ("ops_begin", "<u4")])
OP_DT = np.dtype([("node", "<u4"), ("len", "<u2"), ("op", "u1"), ("pad", "u1")])
WINDOW_DT = np.dtype([("read_off", "<u8"), ("read_len", "<u4"), ("flags", "<u4"), ("first_node", "<u4"), ("n_nodes", "<u4"),
                      ("max_gap_length", "<u4"), ("reserved", "<u4")])
assert WINDOW_DT.itemsize == 32
EXTENSION_DT = np.dtype([("read_off", "<u8"), ("read_len", "<u4"), ("flags", "<u4"), ("first_node", "<u4"), ("n_nodes", "<u4"), ("max_gap_length", "<u4"),
                         ("start_node", "<u4"), ("start_offset", "<u4"), ("query_offset", "<u4"), ("leftward", "<u4"), ("reserved", "<u4")])
assert EXTENSION_DT.itemsize == 48
RESCUE_REQUEST_DT = np.dtype([("mapped", "<u4"), ("lost", "<u4"), ("node_lo", "<u4"), ("node_hi", "<u4"), ("seed_begin", "<i4"), ("seed_end", "<i4"),
                              ("seed_node", "<i4"), ("seed_offset", "<i4"), ("reverse", "<u4"), ("reserved", "<u4")])
assert RESCUE_REQUEST_DT.itemsize == 40
BANDED_DT = np.dtype([("read", "<u8"), ("qual", "<u8"), ("read_len", "<u4"), ("flags", "<u4"), ("graph", GRAPH_DT),
                      ("band_padding", "<i4"), ("reserved", "<u4"), ("max_cells", "<u8")])
VGK_BANDED_PERMISSIVE = 1
SEED_DT = np.dtype([("node", "<u4"), ("diff", "<i4")])
MINIMIZER_HIT_DT = np.dtype([("key", "<u8"), ("node", "<u4"), ("offset", "<u4")])
TAIL_ALIGNMENT_DT = np.dtype([("ext", "<u4"), ("left", "<u4"), ("read_begin", "<u4"), ("read_end", "<u4"), ("score", "<i4"), ("status", "<i4"),
                              ("ops_begin", "<u4"), ("n_ops", "<u4"), ("first_offset", "<u4"), ("n_trees", "<u4")])
GAPLESS_DT = np.dtype([("read", "<u8"), ("read_len", "<u4"), ("n_seeds", "<u4"), ("seeds", "<u8"), ("max_mismatches", "<u4"),
                       ("flags", "<u4"), ("overlap_threshold", "<f8")])
EXT_DT = np.dtype([("path_begin", "<u4"), ("path_len", "<u4"), ("offset", "<u4"), ("read_begin", "<u4"), ("read_end", "<u4"),
                   ("mism_begin", "<u4"), ("n_mismatches", "<u4"), ("score", "<i4"), ("left_full", "u1"), ("right_full", "u1"),
                   ("pad", "u1", 2), ("state", "<u4", 6)])
GAPLESS_RESULT_DT = np.dtype([("status", "<i4"), ("ext_begin", "<u4"), ("n_ext", "<u4"), ("full_length", "<u4")])
VGK_GAPLESS_TRIM = 1
VGK_GAPLESS_DEFER = 2
WFA_DT = np.dtype([("seq", "<u8"), ("seq_len", "<u4"), ("mode", "<u4"), ("from_node", "<u4"), ("from_offset", "<u4"),
                   ("to_node", "<u4"), ("to_offset", "<u4")])
WFA_RESULT_DT = np.dtype([("status", "<i4"), ("ok", "<i4"), ("score", "<i4"), ("node_offset", "<u4"), ("seq_offset", "<u4"),
                          ("length", "<u4"), ("path_begin", "<u4"), ("path_len", "<u4"), ("edit_begin", "<u4"), ("n_edits", "<u4")])
WFA_EVENT_DT = np.dtype([("per_base", "<f8"), ("min", "<i4"), ("max", "<i4")])
WFA_CONNECT, WFA_SUFFIX, WFA_PREFIX = 0, 1, 2
WFA_MATCH, WFA_MISMATCH, WFA_INSERTION, WFA_DELETION = 0, 1, 2, 3
WFA_NO_NODE = 0xffffffff
WFA_DEFAULT_MODEL = ((0.03, 1, 6), (0.05, 1, 10), (0.1, 1, 20), (0.1, 10, 200))      # gbwt_extender.hpp:386-395
READ_MINIMIZER_DT = np.dtype([("key", "<u8"), ("offset", "<u4"), ("hits", "<u4"), ("flags", "<u4"), ("reserved", "<u4")])      # vgk_read_minimizer
MINIMIZER_REVERSE = 1
# vgk_chain_stitch (include/vgk.h): pieces of a read's chain in, one composed alignment per read out
CHAIN_PIECE_DT = np.dtype([("kind", "<u4"), ("link", "<u4"), ("node_offset", "<u4"), ("path_begin", "<u4"), ("path_len", "<u4"), ("edit_begin", "<u4"), ("n_edits", "<u4"), ("reserved", "<u4")])
CHAIN_MAPPING_DT = np.dtype([("node", "<u4"), ("offset", "<u4"), ("edit_begin", "<u4"), ("n_edits", "<u4")])
CHAIN_RESULT_DT = np.dtype([("status", "<i4"), ("mapping_begin", "<u4"), ("n_mappings", "<u4"), ("edit_begin", "<u4"), ("n_edits", "<u4"), ("from_length", "<u4"), ("to_length", "<u4"), ("reserved", "<u4")])
PIECE_LINK, PIECE_ALIGNMENT, PIECE_PATH = 0, 1, 2
assert WFA_DT.itemsize == 32 and WFA_RESULT_DT.itemsize == 40 and WFA_EVENT_DT.itemsize == 16
assert CHAIN_PIECE_DT.itemsize == 32 and CHAIN_MAPPING_DT.itemsize == 16 and CHAIN_RESULT_DT.itemsize == 32
assert GAPLESS_DT.itemsize == 40 and EXT_DT.itemsize == 60 and GAPLESS_RESULT_DT.itemsize == 16
assert BANDED_DT.itemsize == 80
assert GRAPH_DT.itemsize == 40 and PROBLEM_DT.itemsize == 80 and RESULT_DT.itemsize == 32 and OP_DT.itemsize == 8


class Scoring(ctypes.Structure):
    _fields_ = [("matrix", ctypes.c_int8 * 25), ("gap_open", ctypes.c_uint8), ("gap_extend", ctypes.c_uint8),
                ("full_length_bonus", ctypes.c_int8), ("reserved", ctypes.c_uint8)]

    @classmethod
    def simple(cls, match=1, mismatch=4, gap_open=6, gap_extend=1, bonus=5):
        """match/mismatch matrix with the all-zero N row/column (src/alignment_scorer.cpp:297-304)."""
        s = cls()
        for i in range(25):
            r, c = divmod(i, 5)
            s.matrix[i] = 0 if (r == 4 or c == 4) else (match if r == c else -mismatch)
        s.gap_open, s.gap_extend, s.full_length_bonus = gap_open, gap_extend, bonus
        return s


class Haplotypes(ctypes.Structure):
    _fields_ = [("n_nodes", ctypes.c_uint32), ("node_len", ctypes.c_void_p), ("seq", ctypes.c_void_p),
                ("n_threads", ctypes.c_uint32), ("thread_off", ctypes.c_void_p), ("thread_nodes", ctypes.c_void_p)]


class QualAdj(ctypes.Structure):
    _fields_ = [("matrix", ctypes.c_void_p), ("bonuses", ctypes.c_void_p)]


class VgkError(RuntimeError):
    pass


def load_library(path=None):
    path = path or os.environ.get("VGAMD_ENGINE_LIB") or DEFAULT_LIB
    if not os.path.exists(path):
        raise VgkError("engine library %s is missing: run `make lib` (python -c 'import __graft_entry__ as g; g.build()'); "
                       "there is no CPU fallback" % path)
    lib = ctypes.CDLL(path)
    vp, u32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t
    lib.vgk_abi_version.restype = ctypes.c_int
    lib.vgk_strerror.restype = ctypes.c_char_p
    lib.vgk_strerror.argtypes = [ctypes.c_int]
    lib.vgk_create.argtypes = [ctypes.c_int, ctypes.POINTER(Scoring), ctypes.POINTER(vp)]
    lib.vgk_create_qual_adj.argtypes = [ctypes.c_int, ctypes.POINTER(Scoring), ctypes.POINTER(QualAdj), ctypes.POINTER(vp)]
    lib.vgk_destroy.argtypes = [vp]
    lib.vgk_device_info.argtypes = [vp, ctypes.c_char_p, sz, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(sz)]
    lib.vgk_gssw_pack.argtypes = [vp, vp, u32, u32, ctypes.POINTER(vp)]
    lib.vgk_gssw_run.argtypes = [vp]
    lib.vgk_gssw_fetch.argtypes = [vp, vp, vp, sz, ctypes.POINTER(sz)]
    lib.vgk_gssw_align.argtypes = [vp, vp, u32, vp, vp, sz, ctypes.POINTER(sz)]
    lib.vgk_batch_free.argtypes = [vp]
    lib.vgk_xdrop_band_align.argtypes = [vp, vp, u32, vp, vp, sz, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_uint64 * 2)]
    lib.vgk_graph_create.argtypes = [vp, vp, ctypes.POINTER(vp)]
    lib.vgk_graph_destroy.argtypes = [vp]
    lib.vgk_gssw_pack_windows.argtypes = [vp, vp, vp, sz, vp, u32, u32, ctypes.POINTER(vp)]
    lib.vgk_batch_sync.argtypes = [vp]
    lib.vgk_batch_kernel_ms.restype = ctypes.c_double
    lib.vgk_batch_kernel_ms.argtypes = [vp, ctypes.c_int]
    lib.vgk_banded_align.argtypes = [vp, vp, u32, vp, vp, sz, ctypes.POINTER(sz)]
    lib.vgk_haplo_create.argtypes = [vp, ctypes.POINTER(Haplotypes), ctypes.POINTER(vp)]
    lib.vgk_haplo_destroy.argtypes = [vp]
    lib.vgk_gapless_extend.argtypes = [vp, vp, vp, u32, vp, vp, sz, vp, sz, vp, sz, ctypes.POINTER(sz * 3)]
    lib.vgk_gapless_last_ms.restype = ctypes.c_double
    lib.vgk_gapless_last_ms.argtypes = [vp]
    lib.vgk_banded_align_multi.argtypes = [vp, vp, u32, u32, vp, vp, vp, sz, ctypes.POINTER(sz)]
    lib.vgk_gssw_align_multi.argtypes = [vp, vp, u32, u32, vp, vp, vp, sz, ctypes.POINTER(sz)]
    lib.vgk_gssw_multi_host_walks.argtypes = [vp]; lib.vgk_gssw_multi_host_walks.restype = ctypes.c_uint64
    lib.vgk_banded_rerun.argtypes = [vp]
    lib.vgk_gapless_rerun.argtypes = [vp]
    lib.vgk_wfa_extend.argtypes = [vp, vp, vp, vp, u32, vp, vp, sz, vp, sz, ctypes.POINTER(sz * 2)]
    lib.vgk_chain_stitch.argtypes = [vp, vp, vp, vp, u32, vp, sz, vp, sz, vp, sz, vp, vp, sz, vp, sz, ctypes.POINTER(sz * 2)]
    lib.vgk_chain_stitch_last_ms.restype = ctypes.c_double; lib.vgk_chain_stitch_last_ms.argtypes = [vp]
    lib.vgk_wfa_rerun.argtypes = [vp]
    lib.vgk_wfa_last_ms.restype = ctypes.c_double
    lib.vgk_wfa_last_ms.argtypes = [vp]
    lib.vgk_banded_last.restype = ctypes.c_double
    lib.vgk_banded_last.argtypes = [vp, ctypes.c_int]
    for f in ("vgk_batch_cells", "vgk_batch_alg_bytes", "vgk_batch_device_bytes", "vgk_batch_wave_steps"):
        getattr(lib, f).restype = ctypes.c_uint64
        getattr(lib, f).argtypes = [vp]
    return lib


class ProblemSet:
    """A batch of gssw problems laid out in shared numpy arenas.

    reads     : uint8 array of all read bases (ASCII), read i = reads[read_off[i]:read_off[i+1]]
    node_len  : uint32, all graphs' node lengths concatenated; graph i owns node_off[i]:node_off[i+1]
    seq       : uint8 ASCII graph bases, graph i's nodes concatenated starting at seq_off[i]
    pred_off  : uint32 per graph CSR offsets (n_nodes+1 entries each, LOCAL to the graph), laid out at node_off[i]+i
    pred_idx  : uint32 predecessor indices, graph i's edges start at edge_off[i]
    """

    def __init__(self, reads, read_off, node_len, node_off, seq, seq_off, pred_off, pred_idx, edge_off, flags, pinning=None,
                 max_gap=None, quals=None):
        self.reads = np.ascontiguousarray(reads, dtype=np.uint8)
        self.read_off = np.asarray(read_off, dtype=np.int64)
        self.node_len = np.ascontiguousarray(node_len, dtype=np.uint32)
        self.node_off = np.asarray(node_off, dtype=np.int64)
        self.seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self.seq_off = np.asarray(seq_off, dtype=np.int64)
        self.pred_off = np.ascontiguousarray(pred_off, dtype=np.uint32)
        self.pred_idx = np.ascontiguousarray(pred_idx if len(pred_idx) else np.zeros(1, np.uint32), dtype=np.uint32)
        self.edge_off = np.asarray(edge_off, dtype=np.int64)
        self.flags = np.asarray(flags, dtype=np.uint32)
        self.pinning = None if pinning is None else np.ascontiguousarray(pinning, dtype=np.uint8)
        self.n = len(self.read_off) - 1
        n = self.n
        arr = np.zeros(n, dtype=PROBLEM_DT)
        arr["read"] = self.reads.ctypes.data + self.read_off[:-1]
        arr["read_len"] = np.diff(self.read_off)
        arr["flags"] = self.flags
        g = arr["graph"]
        g["n_nodes"] = np.diff(self.node_off)
        g["node_len"] = self.node_len.ctypes.data + 4 * self.node_off[:-1]
        g["seq"] = self.seq.ctypes.data + self.seq_off[:-1]
        g["pred_off"] = self.pred_off.ctypes.data + 4 * (self.node_off[:-1] + np.arange(n))
        g["pred_idx"] = self.pred_idx.ctypes.data + 4 * self.edge_off[:-1]
        arr["graph"] = g
        if self.pinning is not None:
            arr["pinning"] = self.pinning.ctypes.data + self.node_off[:-1]
        if max_gap is not None:
            arr["max_gap_length"] = np.asarray(max_gap, dtype=np.uint32)
        self.quals = None if quals is None else np.ascontiguousarray(quals, dtype=np.uint8)   # raw phred, same layout as reads
        if self.quals is not None:
            arr["qual"] = self.quals.ctypes.data + self.read_off[:-1]
        self.array = arr

    @property
    def ptr(self):
        return self.array.ctypes.data

    def subset(self, k):
        """First k problems (same arenas, no copies)."""
        s = object.__new__(ProblemSet)
        s.__dict__.update(self.__dict__)
        s.n = k; s.array = self.array[:k]; s.read_off = self.read_off[:k + 1]; s.seq_off = self.seq_off[:k + 1]
        return s

    @classmethod
    def from_lists(cls, problems):
        """problems: list of dicts {read: str, nodes: [str], preds: [[int]], flags: int, pinning: [0/1]|None}."""
        reads, read_off, node_len, node_off, seq, seq_off, pred_off, pred_idx, edge_off, flags, pinning = \
            [], [0], [], [0], [], [0], [], [], [0], [], []
        any_pin = any(p.get("pinning") is not None for p in problems)
        for p in problems:
            reads.append(np.frombuffer(p["read"].encode(), dtype=np.uint8)); read_off.append(read_off[-1] + len(p["read"]))
            nl = [len(s) for s in p["nodes"]]
            node_len.extend(nl); node_off.append(node_off[-1] + len(nl))
            s = "".join(p["nodes"]); seq.append(np.frombuffer(s.encode(), dtype=np.uint8)); seq_off.append(seq_off[-1] + len(s))
            off = [0]
            for pr in p["preds"]:
                pred_idx.extend(pr); off.append(off[-1] + len(pr))
            pred_off.extend(off); edge_off.append(edge_off[-1] + off[-1])
            flags.append(p["flags"])
            pinning.extend(p["pinning"] if p.get("pinning") is not None else [0] * len(nl))
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.uint8)
        quals = None
        if any(p.get("qual") is not None for p in problems):
            quals = np.concatenate([np.asarray(p["qual"], dtype=np.uint8) for p in problems])
        return cls(cat(reads), read_off, node_len, node_off, cat(seq), seq_off, pred_off, pred_idx, edge_off, flags,
                   pinning if any_pin else None, [p.get("max_gap", 40) for p in problems], quals)


class BandedSet(ProblemSet):
    """A batch of banded-global problems (vgk_banded_problem) over the same arenas as ProblemSet.
    problems: list of dicts {read, nodes: [str] (may be empty strings), preds, band_padding, permissive, max_cells?, qual?}."""

    @classmethod
    def from_lists(cls, problems):
        base = ProblemSet.from_lists([dict(p, flags=0, pinning=None) for p in problems])
        s = object.__new__(cls)
        s.__dict__.update(base.__dict__)
        arr = np.zeros(s.n, dtype=BANDED_DT)
        for k in ("read", "read_len", "graph"):
            arr[k] = base.array[k]
        arr["qual"] = base.array["qual"]
        arr["flags"] = [VGK_BANDED_PERMISSIVE if p.get("permissive", True) else 0 for p in problems]
        arr["band_padding"] = [p.get("band_padding", 1) for p in problems]
        arr["max_cells"] = [p.get("max_cells", 0) for p in problems]
        s.array = arr
        return s

    def subset(self, k):
        raise NotImplementedError

    def select(self, idx):
        """the problems idx of this set as a set of their own over the SAME arenas (only the problem structs are copied): a caller that
        keeps its extracted subgraphs flat picks a batch out of them without touching a base"""
        idx = np.asarray(idx, dtype=np.int64)
        s = object.__new__(type(self))
        s.__dict__.update(self.__dict__)
        s._arenas_of = self                                         # (keeps the arenas alive)
        s.array = np.ascontiguousarray(self.array[idx]); s.n = len(idx)
        rl = np.diff(self.read_off)[idx]; sl = np.diff(self.seq_off)[idx]; nn = self.array["graph"]["n_nodes"][idx].astype(np.int64)
        s.ops_cap = int(rl.sum() + sl.sum() + 2 * nn.sum() + 8 * s.n)
        return s


class Engine:
    """One engine context = one (device, scoring) pair, like one vg Aligner."""

    def __init__(self, scoring=None, device=0, lib=None, qual_adj=None):
        """qual_adj = (matrix int8[256*25], bonuses int8[256]) makes a quality-adjusted context (QualAdjAligner)."""
        self.lib = load_library(lib) if (lib is None or isinstance(lib, str)) else lib
        self.scoring = scoring or Scoring.simple()
        self.device = device
        h = ctypes.c_void_p()
        if qual_adj is not None:
            self._qm = np.ascontiguousarray(qual_adj[0], dtype=np.int8); self._qb = np.ascontiguousarray(qual_adj[1], dtype=np.int8)
            assert self._qm.size == 256 * 25 and self._qb.size == 256
            qa = QualAdj(self._qm.ctypes.data, self._qb.ctypes.data)
            rc = self.lib.vgk_create_qual_adj(device, ctypes.byref(self.scoring), ctypes.byref(qa), ctypes.byref(h))
        else:
            rc = self.lib.vgk_create(device, ctypes.byref(self.scoring), ctypes.byref(h))
        if rc != VGK_OK:
            raise VgkError("vgk_create: %s" % self.lib.vgk_strerror(rc).decode())
        self.h = h
        self._indexes = weakref.WeakSet()      # haplotype indexes live in this context: they go first

    def set_speculation(self, mode):
        """0 = by feedback (default), 1 = whenever a batch allows it, 2 = never (vgk_set_speculation)"""
        self.lib.vgk_set_speculation.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self._check(self.lib.vgk_set_speculation(self.h, mode), "vgk_set_speculation")

    def speculation_state(self):
        """-> dict(on, observed, turned_off, turned_on, probe_interval, last_miss): the context's feedback on the speculative fill"""
        self.lib.vgk_speculation_state.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        c = (ctypes.c_uint64 * 4)(); m = ctypes.c_double()
        on = self.lib.vgk_speculation_state(self.h, c, ctypes.byref(m))
        return dict(on=bool(on), observed=int(c[0]), turned_off=int(c[1]), turned_on=int(c[2]), probe_interval=int(c[3]), last_miss=float(m.value))

    def close(self):
        if getattr(self, "h", None):
            for index in list(getattr(self, "_indexes", ())):
                index.close()
            self.lib.vgk_destroy(self.h); self.h = None

    __del__ = close

    def host_register(self, array):
        """vgk_host_register: page-lock a numpy array this caller keeps (copies out of it then run at the link's rate); False when refused"""
        a = np.ascontiguousarray(array)
        self.lib.vgk_host_register.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        return self.lib.vgk_host_register(self.h, a.ctypes.data, a.nbytes) == VGK_OK

    def host_unregister(self, array):
        self.lib.vgk_host_unregister.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        return self.lib.vgk_host_unregister(self.h, np.ascontiguousarray(array).ctypes.data) == VGK_OK

    def device_info(self):
        name = ctypes.create_string_buffer(256); cus = ctypes.c_int(); mem = ctypes.c_size_t()
        self.lib.vgk_device_info(self.h, name, 256, ctypes.byref(cus), ctypes.byref(mem))
        return name.value.decode(), cus.value, mem.value

    def _check(self, rc, what):
        if rc != VGK_OK:
            raise VgkError("%s: %s" % (what, self.lib.vgk_strerror(rc).decode()))

    def pack(self, ps, ops_per_problem=0):
        b = ctypes.c_void_p()
        self._check(self.lib.vgk_gssw_pack(self.h, ps.ptr, ps.n, ops_per_problem, ctypes.byref(b)), "vgk_gssw_pack")
        return Batch(self, b, ps, ops_per_problem)

    def graph(self, node_len, seq, pred_off, pred_idx):
        """One DAG (nodes in topological order, predecessor CSR) resident in HBM -> ResidentGraph."""
        return ResidentGraph(self, node_len, seq, pred_off, pred_idx)

    def pack_windows(self, graph, ws, ops_per_problem=0):
        """vgk_gssw_pack_windows: ws = WindowSet (reads + windows of `graph`); packed on the device."""
        b = ctypes.c_void_p()
        self._check(self.lib.vgk_gssw_pack_windows(self.h, graph.h, ws.reads.ctypes.data, ws.reads.size, ws.array.ctypes.data, ws.n,
                                                   ops_per_problem, ctypes.byref(b)), "vgk_gssw_pack_windows")
        return Batch(self, b, ws, ops_per_problem)

    def pack_extensions(self, graph, es, ops_per_problem=0):
        """vgk_gssw_pack_extensions: es = ExtensionSet; the sub-DAGs are derived on the device"""
        b = ctypes.c_void_p()
        self.lib.vgk_gssw_pack_extensions.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        self._check(self.lib.vgk_gssw_pack_extensions(self.h, graph.h, es.reads.ctypes.data, es.reads.size, es.array.ctypes.data, es.n,
                                                      ops_per_problem, ctypes.byref(b)), "vgk_gssw_pack_extensions")
        return Batch(self, b, es, ops_per_problem)

    def rescue_requests(self, graph, mean, sd, stdevs=4.0, out=None):
        """vgk_rescue_requests over the sets the last gapless_extend(_seeded) call on this context left in HBM; graph: a DeviceGraph or a vgk_dgraph
        pointer (int) of the same device; out: a RESCUE_REQUEST_DT array to fill (grown when too small) -> the filled part"""
        gh = graph if isinstance(graph, int) else graph.h
        self.lib.vgk_rescue_requests.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        if out is None:
            out = np.zeros(4096, dtype=RESCUE_REQUEST_DT)
        w = ctypes.c_size_t()
        rc = self.lib.vgk_rescue_requests(self.h, gh, float(mean), float(sd), float(stdevs), out.ctypes.data, len(out), ctypes.byref(w))
        if rc == VGK_EOPS:
            out = np.zeros(int(w.value) + int(w.value) // 8 + 64, dtype=RESCUE_REQUEST_DT)
            rc = self.lib.vgk_rescue_requests(self.h, gh, float(mean), float(sd), float(stdevs), out.ctypes.data, len(out), ctypes.byref(w))
        self._check(rc, "vgk_rescue_requests")
        self._rescue_requests_buf = out
        return out[:int(w.value)]

    def align_extensions(self, graph, es, ops_per_problem=0):
        with self.pack_extensions(graph, es, ops_per_problem) as b:
            b.run()
            return b.fetch()

    def align_windows(self, graph, ws, ops_per_problem=0):
        with self.pack_windows(graph, ws, ops_per_problem) as b:
            b.run()
            return b.fetch()

    def align(self, ps, ops_per_problem=0):
        with self.pack(ps, ops_per_problem) as b:
            b.run()
            return b.fetch()

    def align_call(self, ps):
        """vgk_gssw_align (the one-call form: sub-batches, and the wide route for problems outside the packed kernels' range) -> (results, ops)."""
        res = np.zeros(ps.n, dtype=RESULT_DT)
        cap = int(np.diff(ps.read_off).sum() + np.diff(ps.seq_off).sum() + 4 * ps.n)
        ops = np.zeros(max(cap, 1), dtype=OP_DT)
        written = ctypes.c_size_t()
        self._check(self.lib.vgk_gssw_align(self.h, ps.ptr, ps.n, res.ctypes.data, ops.ctypes.data, cap, ctypes.byref(written)), "vgk_gssw_align")
        return res, ops[:written.value]

    def xdrop_band_align(self, ps, out=None):
        """vgk_xdrop_band_align over a ProblemSet of VGK_XDROP_PINNED problems -> (results, ops, (cells in band, cells of the rectangles)).
        out: (results, ops) arrays of an earlier call on the same set, written again (a caller that keeps its output buffers)."""
        if out is not None:
            res, ops = out; cap = len(ops)
        else:
            res = np.zeros(ps.n, dtype=RESULT_DT)
            cap = int(np.diff(ps.read_off).sum() + np.diff(ps.seq_off).sum() + len(ps.node_len) + 4 * ps.n)
            ops = np.zeros(max(cap, 1), dtype=OP_DT)
        written = ctypes.c_size_t(); stats = (ctypes.c_uint64 * 2)()
        self._check(self.lib.vgk_xdrop_band_align(self.h, ps.ptr, ps.n, res.ctypes.data, ops.ctypes.data, cap, ctypes.byref(written), ctypes.byref(stats)),
                    "vgk_xdrop_band_align")
        return res, ops[:written.value], (int(stats[0]), int(stats[1]))

    def xdrop_band_last_classes(self):
        """problems of the last xdrop_band_align call by the lanes they ran on: (8, 16, 64)"""
        self.lib.vgk_xdrop_band_last_class.restype = ctypes.c_uint64; self.lib.vgk_xdrop_band_last_class.argtypes = [ctypes.c_void_p, ctypes.c_int]
        return tuple(int(self.lib.vgk_xdrop_band_last_class(self.h, k)) for k in range(3))

    def align_multi(self, ps, max_alt_alns):
        """vgk_gssw_align_multi over a ProblemSet of pinned problems -> (results [n, max_alt_alns], n_alignments [n], ops)."""
        res = np.zeros((ps.n, max_alt_alns), dtype=RESULT_DT); cnt = np.zeros(ps.n, dtype=np.uint32)
        cap = int((np.diff(ps.read_off).sum() + np.diff(ps.seq_off).sum() + 4 * ps.n) * max_alt_alns)
        ops = np.zeros(max(cap, 1), dtype=OP_DT)
        written = ctypes.c_size_t()
        self._check(self.lib.vgk_gssw_align_multi(self.h, ps.ptr, ps.n, max_alt_alns, res.ctypes.data, cnt.ctypes.data, ops.ctypes.data, cap,
                                                  ctypes.byref(written)), "vgk_gssw_align_multi")
        self.multi_host_walks = int(self.lib.vgk_gssw_multi_host_walks(self.h))      # problems whose alternates a host thread walked
        return res, cnt, ops[:written.value]

    def banded_align(self, bs):
        """vgk_banded_align over a BandedSet -> (results, ops); per-problem failures are reported in results['status']."""
        res = np.zeros(bs.n, dtype=RESULT_DT)
        cap = getattr(bs, "ops_cap", None)
        if cap is None:
            cap = int(np.diff(bs.read_off).sum() + np.diff(bs.seq_off).sum() + 2 * len(bs.node_len) + 8 * bs.n)
        ops = np.zeros(max(cap, 1), dtype=OP_DT)
        written = ctypes.c_size_t()
        self._check(self.lib.vgk_banded_align(self.h, bs.ptr, bs.n, res.ctypes.data, ops.ctypes.data, cap, ctypes.byref(written)),
                    "vgk_banded_align")
        return res, ops[:written.value]

    def banded_align_multi(self, bs, max_alt_alns):
        """vgk_banded_align_multi -> (results [n, max_alt_alns], n_alignments [n], ops)"""
        res = np.zeros((bs.n, max_alt_alns), dtype=RESULT_DT)
        cnt = np.zeros(bs.n, dtype=np.uint32)
        cap = int((np.diff(bs.read_off).sum() + np.diff(bs.seq_off).sum() + 2 * len(bs.node_len) + 8 * bs.n) * max_alt_alns)
        ops = np.zeros(max(cap, 1), dtype=OP_DT)
        written = ctypes.c_size_t()
        self._check(self.lib.vgk_banded_align_multi(self.h, bs.ptr, bs.n, max_alt_alns, res.ctypes.data, cnt.ctypes.data, ops.ctypes.data, cap,
                                                    ctypes.byref(written)), "vgk_banded_align_multi")
        self.multi_host_walks = int(self.lib.vgk_gssw_multi_host_walks(self.h))      # problems whose alternates a host thread walked
        return res, cnt, ops[:written.value]

    def banded_rerun(self):
        self._check(self.lib.vgk_banded_rerun(self.h), "vgk_banded_rerun")

    def gapless_rerun(self):
        self._check(self.lib.vgk_gapless_rerun(self.h), "vgk_gapless_rerun")

    def banded_last(self, which):
        return self.lib.vgk_banded_last(self.h, which)

    def _out(self, name, n, dtype):
        """an output array of n elements.  reuse_outputs = True (a caller that consumes one call's outputs before the next call, like
        bench.py's steady-state loops — or any C caller with its own buffers): one array per output is kept and grown, so a call neither
        allocates nor page-faults hundreds of MB; the arrays a call returns are then views that the next call overwrites."""
        if not getattr(self, "reuse_outputs", False):
            return np.zeros(n, dtype=dtype)
        cache = self.__dict__.setdefault("_out_cache", {})
        a = cache.get(name)
        if a is None or len(a) < n or a.dtype != dtype:
            a = cache[name] = np.zeros(max(n, 1), dtype=dtype)
        return a[:n]

    def haplo_index(self, nodes, threads):
        """nodes: [str] in node-id order; threads: [[oriented node = 2 * index + is_reverse]].  -> HaploIndex"""
        return HaploIndex(self, nodes, threads)

    def gbz_load(self, gbz_bytes):
        """vgk_gbz_load: a GBZ image -> (node sequences [str], threads [[oriented node]])"""
        lib = self.lib
        out = ctypes.POINTER(Haplotypes)()
        lib.vgk_gbz_load.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
        self._check(lib.vgk_gbz_load(bytes(gbz_bytes), len(gbz_bytes), ctypes.byref(out)), "vgk_gbz_load")
        try:
            h = out.contents
            lens = np.ctypeslib.as_array(ctypes.cast(h.node_len, ctypes.POINTER(ctypes.c_uint32)), (h.n_nodes,)).copy()
            seq = ctypes.string_at(h.seq, int(lens.sum())).decode()
            at = [0] + [int(x) for x in np.cumsum(lens.astype(np.int64))]
            nodes = [seq[at[i]:at[i + 1]] for i in range(h.n_nodes)]
            toff = np.ctypeslib.as_array(ctypes.cast(h.thread_off, ctypes.POINTER(ctypes.c_uint32)), (h.n_threads + 1,)).copy()
            tn = np.ctypeslib.as_array(ctypes.cast(h.thread_nodes, ctypes.POINTER(ctypes.c_uint32)), (max(int(toff[-1]), 1),)).copy()
            threads = [[int(x) for x in tn[toff[t]:toff[t + 1]]] for t in range(h.n_threads)]
        finally:
            lib.vgk_haplotypes_free.argtypes = [ctypes.c_void_p]; lib.vgk_haplotypes_free.restype = None
            lib.vgk_haplotypes_free(out)
        return nodes, threads

    def haplo_index_from_gbwt(self, nodes, gbwt_bytes):
        """vgk_haplo_create_gbwt: the index from the image of a (simple-sds, bidirectional) GBWT file"""
        return HaploIndex(self, nodes, gbwt=gbwt_bytes)

    def gapless_extend(self, index, problems, defer=False):
        """problems: a GaplessSet, or a list of dicts {read, seeds: [(oriented node, read_offset - node_offset)], max_mismatches?,
        overlap_threshold?, trim?}.  -> (results, extensions, nodes, mismatches) as numpy arrays laid out like include/vgk.h."""
        gs = problems if isinstance(problems, GaplessSet) else GaplessSet.from_lists(problems)
        res = self._out("g_res", gs.n, GAPLESS_RESULT_DT)
        ext = self._out("g_ext", gs.ext_cap, EXT_DT); nodes = self._out("g_nodes", gs.node_cap, np.uint32); mism = self._out("g_mism", gs.mism_cap, np.uint32)
        written = (ctypes.c_size_t * 3)()
        if defer and gs.n:                                   # VGK_GAPLESS_DEFER rides on the first problem's flags (filled when the next tail_stage* call returns)
            gs.array["flags"][0] |= VGK_GAPLESS_DEFER
        try:
            self._check(self.lib.vgk_gapless_extend(self.h, index.h, gs.array.ctypes.data, gs.n, res.ctypes.data, ext.ctypes.data, gs.ext_cap,
                                                    nodes.ctypes.data, gs.node_cap, mism.ctypes.data, gs.mism_cap, ctypes.byref(written)),
                        "vgk_gapless_extend")
        finally:
            if defer and gs.n:
                gs.array["flags"][0] &= ~np.uint32(VGK_GAPLESS_DEFER)
        if defer:
            self._deferred_keep = (res, ext, nodes, mism)          # the engine still writes into them: alive until the deferral is finished
        return res, ext[:written[0]], nodes[:written[1]], mism[:written[2]]

    def minimizer_index(self, nodes, threads, k=29, w=11):
        return MinimizerIndex(self, nodes, threads, k, w)

    def minimizer_seeds(self, mindex, hindex, reads, read_off, hit_cap=500, keep_on_device=False):
        """vgk_minimizer_seeds: reads flat (uint8), read i = reads[read_off[i]:read_off[i+1]] -> (seed_off [n+1], seeds as SEED_DT, minimizers per read);
        keep_on_device: the seeds stay in HBM for gapless_extend_seeded (an empty seeds array comes back).  `self.minimizers_truncated`: per
        read, whether it reached the cap of 64 seeds with hits left unexamined (VGK_MINIMIZERS_TRUNCATED)"""
        reads = np.ascontiguousarray(reads, dtype=np.uint8); off = np.ascontiguousarray(read_off, dtype=np.uint64)
        n = len(off) - 1
        seed_off = self._out("mz_off", n + 1, np.uint32); mins = self._out("mz_mins", max(n, 1), np.uint32)
        cap = 0 if keep_on_device else 64 * max(n, 1)
        seeds = self._out("mz_seeds", max(cap, 1), SEED_DT)
        written = ctypes.c_size_t()
        self.lib.vgk_minimizer_seeds.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        self._check(self.lib.vgk_minimizer_seeds(self.h, mindex.h, hindex.h, reads.ctypes.data, off.ctypes.data, n, hit_cap, seed_off.ctypes.data, mins.ctypes.data,
                                                 None if keep_on_device else seeds.ctypes.data, cap, ctypes.byref(written)), "vgk_minimizer_seeds")
        self.minimizers_truncated = (mins[:n] & 0x80000000) != 0
        self.minimizers_policy_skipped = (mins[:n] & 0x40000000) != 0          # (VGK_MINIMIZERS_POLICY_SKIPPED: more than 64 minimizers, seeded without the policy)
        return seed_off, seeds[:0 if keep_on_device else written.value], mins[:n] & 0x3fffffff

    def minimizer_list(self, mindex, reads, read_off):
        """vgk_minimizer_list: every minimizer of every read, no caps -> (minimizer_off [n + 1] uint64, records READ_MINIMIZER_DT in read order)"""
        reads = np.ascontiguousarray(reads, dtype=np.uint8); off = np.ascontiguousarray(read_off, dtype=np.uint64); n = len(off) - 1
        moff = np.zeros(n + 1, dtype=np.uint64); written = ctypes.c_size_t()
        self.lib.vgk_minimizer_list.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        cap = max(16, int(len(reads)) // 4)
        for attempt in range(2):
            recs = np.zeros(cap, dtype=READ_MINIMIZER_DT)
            rc = self.lib.vgk_minimizer_list(self.h, mindex.h, reads.ctypes.data, off.ctypes.data, n, moff.ctypes.data, recs.ctypes.data, cap, ctypes.byref(written))
            if rc != VGK_EOPS:
                break
            cap = int(written.value)
        self._check(rc, "vgk_minimizer_list")
        return moff, recs[:written.value]

    def minimizer_seeds_of(self, mindex, minimizers, take):
        """vgk_minimizer_seeds_of: one seed per hit of the minimizers taken -> (seed_off [len(minimizers) + 1] uint64, seeds SEED_DT)"""
        recs = np.ascontiguousarray(minimizers, dtype=READ_MINIMIZER_DT); take = np.ascontiguousarray(take, dtype=np.uint8)
        assert len(take) == len(recs)
        soff = np.zeros(len(recs) + 1, dtype=np.uint64); written = ctypes.c_size_t()
        self.lib.vgk_minimizer_seeds_of.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        cap = max(16, int(recs["hits"][take != 0].sum()))
        seeds = np.zeros(cap, dtype=SEED_DT)
        self._check(self.lib.vgk_minimizer_seeds_of(self.h, mindex.h, recs.ctypes.data, take.ctypes.data, len(recs), soff.ctypes.data, seeds.ctypes.data, cap, ctypes.byref(written)), "vgk_minimizer_seeds_of")
        return soff, seeds[:written.value]

    def minimizer_last_ms(self):
        self.lib.vgk_minimizer_last_ms.restype = ctypes.c_double; self.lib.vgk_minimizer_last_ms.argtypes = [ctypes.c_void_p]
        return self.lib.vgk_minimizer_last_ms(self.h)

    def tail_forest(self, index, problems):
        """vgk_tail_forest: problems = numpy array of TAIL_DT (search state node / lo / hi, cut offset, walk distance) or a list of
        such tuples.  -> (results as TAIL_RESULT_DT, Forest)"""
        pr = np.ascontiguousarray(problems, dtype=TAIL_DT) if isinstance(problems, np.ndarray) else np.array([tuple(p) for p in problems], dtype=TAIL_DT)
        res = np.zeros(max(len(pr), 1), dtype=TAIL_RESULT_DT)
        h = ctypes.c_void_p()
        self.lib.vgk_tail_forest.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        self._check(self.lib.vgk_tail_forest(self.h, index.h, pr.ctypes.data, len(pr), res.ctypes.data, ctypes.byref(h)), "vgk_tail_forest")
        return res[:len(pr)], Forest(self, h)

    def tail_stage(self, index, n_reads, n_ext, ops_per_problem=32):
        """vgk_tail_stage over the sets the last gapless_extend / gapless_extend_seeded call left on the device
        -> (ext_total [n_ext], read_score [n_reads], (tails, trees, tree nodes, declined))"""
        ext_total = self._out("ts_ext", max(n_ext, 1), np.int32); read_score = self._out("ts_read", max(n_reads, 1), np.int32); stats = np.zeros(4, dtype=np.uint64)
        self.lib.vgk_tail_stage.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        self._check(self.lib.vgk_tail_stage(self.h, index.h, ops_per_problem, ext_total.ctypes.data, len(ext_total), read_score.ctypes.data, stats.ctypes.data), "vgk_tail_stage")
        return ext_total[:n_ext], read_score[:n_reads], tuple(int(x) for x in stats)

    def tail_stage_aligned(self, index, n_reads, n_ext, ops_per_problem=32):
        """vgk_tail_stage_aligned -> (ext_total, read_score, tails as TAIL_ALIGNMENT_DT, ops as OP_DT, stats)"""
        ext_total = self._out("ts_ext", max(n_ext, 1), np.int32); read_score = self._out("ts_read", max(n_reads, 1), np.int32); stats = np.zeros(4, dtype=np.uint64)
        tails_cap = 2 * max(n_ext, 1)
        tails = self._out("ts_tails", tails_cap, TAIL_ALIGNMENT_DT); ops = self._out("ts_ops", tails_cap * ops_per_problem + 1, OP_DT)
        written = (ctypes.c_size_t * 2)()
        self.lib.vgk_tail_stage_aligned.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                    ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        self._check(self.lib.vgk_tail_stage_aligned(self.h, index.h, ops_per_problem, ext_total.ctypes.data, len(ext_total), read_score.ctypes.data,
                                                    tails.ctypes.data, tails_cap, ops.ctypes.data, len(ops), ctypes.byref(written), stats.ctypes.data), "vgk_tail_stage_aligned")
        return ext_total[:n_ext], read_score[:n_reads], tails[:written[0]], ops[:written[1]], tuple(int(x) for x in stats)

    def tail_stage_last_ms(self):
        self.lib.vgk_tail_stage_last_ms.restype = ctypes.c_double; self.lib.vgk_tail_stage_last_ms.argtypes = [ctypes.c_void_p, ctypes.c_int]
        return [self.lib.vgk_tail_stage_last_ms(self.h, k) for k in range(4)]

    def tail_last_ms(self):
        self.lib.vgk_tail_last_ms.restype = ctypes.c_double; self.lib.vgk_tail_last_ms.argtypes = [ctypes.c_void_p]
        return self.lib.vgk_tail_last_ms(self.h)

    def gapless_extend_seeded(self, index, n_reads, n_seeds, max_mismatches=4, overlap_threshold=0.8, trim=True, read_len=150, defer=False):
        """vgk_gapless_extend_seeded: extend the clusters the last minimizer_seeds call left on the device -> as gapless_extend.
        defer (VGK_GAPLESS_DEFER): the arrays come back sized but are FILLED only when the next tail_stage / tail_stage_aligned call (or
        gapless_fetch_deferred) returns"""
        res = self._out("gs_res", max(n_reads, 1), GAPLESS_RESULT_DT)
        ext_cap = n_seeds + 1; node_cap = n_seeds * 16 + 1024; mism_cap = n_seeds * 12 + 1024
        ext = self._out("gs_ext", ext_cap, EXT_DT); nodes = self._out("gs_nodes", node_cap, np.uint32); mism = self._out("gs_mism", mism_cap, np.uint32)
        written = (ctypes.c_size_t * 3)()
        self.lib.vgk_gapless_extend_seeded.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_double, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                                       ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        self._check(self.lib.vgk_gapless_extend_seeded(self.h, index.h, max_mismatches, overlap_threshold, (VGK_GAPLESS_TRIM if trim else 0) | (VGK_GAPLESS_DEFER if defer else 0), res.ctypes.data,
                                                       ext.ctypes.data, ext_cap, nodes.ctypes.data, node_cap, mism.ctypes.data, mism_cap, ctypes.byref(written)),
                    "vgk_gapless_extend_seeded")
        if defer:
            self._deferred_keep = (res, ext, nodes, mism)          # the engine still writes into them: alive until the deferral is finished
        return res[:n_reads], ext[:written[0]], nodes[:written[1]], mism[:written[2]]

    def gapless_fetch_deferred(self):
        self.lib.vgk_gapless_fetch_deferred.argtypes = [ctypes.c_void_p]
        self._check(self.lib.vgk_gapless_fetch_deferred(self.h), "vgk_gapless_fetch_deferred")

    def gapless_last_ms(self):
        return self.lib.vgk_gapless_last_ms(self.h)

    def gapless_last_redone(self):
        """seeds of the last extension call whose search branched on the merged-run index and ran again on the original one"""
        self.lib.vgk_gapless_last_redone.restype = ctypes.c_uint64; self.lib.vgk_gapless_last_redone.argtypes = [ctypes.c_void_p]
        return int(self.lib.vgk_gapless_last_redone(self.h))

    def gapless_last_retried(self):
        self.lib.vgk_gapless_last_retried.restype = ctypes.c_uint64
        self.lib.vgk_gapless_last_retried.argtypes = [ctypes.c_void_p]
        return self.lib.vgk_gapless_last_retried(self.h)

    def wfa_extend(self, index, problems, error_model=None):
        """problems: a WfaSet, or a list of dicts {seq, mode: "connect"|"suffix"|"prefix", from: (oriented node, offset),
        to: (oriented node, offset)}; error_model: four (per_base, min, max) rows or None for the reference's default.
        -> (results, paths, edits) as numpy arrays laid out like include/vgk.h."""
        ws = problems if isinstance(problems, WfaSet) else WfaSet.from_lists(problems)
        res = np.zeros(ws.n, dtype=WFA_RESULT_DT)
        paths = np.zeros(ws.path_cap, dtype=np.uint32); edits = np.zeros(ws.edit_cap, dtype=np.uint32)
        model = None
        if error_model is not None:
            model = np.zeros(4, dtype=WFA_EVENT_DT)
            for i, row in enumerate(error_model):
                model[i] = tuple(row)
        written = (ctypes.c_size_t * 2)()
        self._check(self.lib.vgk_wfa_extend(self.h, index.h, model.ctypes.data if model is not None else None, ws.array.ctypes.data, ws.n,
                                            res.ctypes.data, paths.ctypes.data, ws.path_cap, edits.ctypes.data, ws.edit_cap,
                                            ctypes.byref(written)), "vgk_wfa_extend")
        return res, paths[:written[0]], edits[:written[1]]

    def chain_stitch(self, index, pieces, piece_off, nodes=None, mappings=None, edits=None, mapping_cap=None, edit_cap=None):
        """vgk_chain_stitch: pieces (CHAIN_PIECE_DT; LINK pieces name results of the last wfa_extend on this engine with `index`), piece_off (n_reads + 1),
        the arrays ALIGNMENT / PATH pieces point into -> (results CHAIN_RESULT_DT, mappings CHAIN_MAPPING_DT, edits uint32: length << 2 | WFA_*).
        Without caps the call is made twice when the first guess is too small."""
        pieces = np.ascontiguousarray(pieces, dtype=CHAIN_PIECE_DT); piece_off = np.ascontiguousarray(piece_off, dtype=np.uint64)
        nodes = np.ascontiguousarray(nodes if nodes is not None else [], dtype=np.uint32)
        mappings = np.ascontiguousarray(mappings if mappings is not None else [], dtype=CHAIN_MAPPING_DT)
        edits = np.ascontiguousarray(edits if edits is not None else [], dtype=np.uint32)
        n = len(piece_off) - 1
        res = np.zeros(n, dtype=CHAIN_RESULT_DT)
        retry = mapping_cap is None and edit_cap is None
        mcap = mapping_cap if mapping_cap is not None else 4 * len(pieces) + len(nodes) + len(mappings) + 16
        ecap = edit_cap if edit_cap is not None else 4 * len(pieces) + len(nodes) + len(edits) + 16
        while True:
            om = np.zeros(mcap, dtype=CHAIN_MAPPING_DT); oe = np.zeros(ecap, dtype=np.uint32)
            written = (ctypes.c_size_t * 2)()
            rc = self.lib.vgk_chain_stitch(self.h, index.h, pieces.ctypes.data if len(pieces) else None, piece_off.ctypes.data, n, nodes.ctypes.data if len(nodes) else None, len(nodes),
                                           mappings.ctypes.data if len(mappings) else None, len(mappings), edits.ctypes.data if len(edits) else None, len(edits),
                                           res.ctypes.data, om.ctypes.data, mcap, oe.ctypes.data, ecap, ctypes.byref(written))
            if rc == VGK_EOPS and retry:
                mcap, ecap, retry = int(written[0]) + 1, int(written[1]) + 1, False
                continue
            if rc != VGK_EOPS:
                self._check(rc, "vgk_chain_stitch")
            return res, om[:min(written[0], mcap)], oe[:min(written[1], ecap)]

    def wfa_set_cost_hints(self, extra_bases):
        """vgk_wfa_set_cost_hints: per problem of the NEXT wfa_extend, bases to add to its length when the hand-out order is made (order only)"""
        a = np.ascontiguousarray(extra_bases, dtype=np.uint32)
        self.lib.vgk_wfa_set_cost_hints.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        self._check(self.lib.vgk_wfa_set_cost_hints(self.h, a.ctypes.data, len(a)), "vgk_wfa_set_cost_hints")

    def wfa_set_point_budgets(self, connect_points, tail_points):
        self.lib.vgk_wfa_set_point_budgets.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
        self._check(self.lib.vgk_wfa_set_point_budgets(self.h, connect_points, tail_points), "vgk_wfa_set_point_budgets")

    def wfa_set_point_budget(self, points):
        """problems that store more than `points` wavefront points are declined (VGK_ETOOBIG) early; 0 = the kernel's table size"""
        self.lib.vgk_wfa_set_point_budget.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        self._check(self.lib.vgk_wfa_set_point_budget(self.h, points), "vgk_wfa_set_point_budget")

    def wfa_rerun(self):
        self._check(self.lib.vgk_wfa_rerun(self.h), "vgk_wfa_rerun")

    def wfa_last_ms(self):
        return self.lib.vgk_wfa_last_ms(self.h)

    def wfa_last_wave(self):
        """(ms of the first launch — the hybrid's pair when its kernels ran at once —, ms of the wavefront kernel when it followed the thread
        kernel, problems handed over / outgrown)"""
        self.lib.vgk_wfa_last_wave.restype = ctypes.c_double; self.lib.vgk_wfa_last_wave.argtypes = [ctypes.c_void_p, ctypes.c_int]
        return tuple(self.lib.vgk_wfa_last_wave(self.h, k) for k in range(3))


TAIL_DT = np.dtype([("node", "<u4"), ("lo", "<i4"), ("hi", "<i4"), ("offset", "<u4"), ("walk_distance", "<u4")])
TAIL_RESULT_DT = np.dtype([("status", "<i4"), ("first_node", "<u4"), ("n_nodes", "<u4"), ("n_trees", "<u4"), ("root_trim", "<u4"), ("bases", "<u4")])


class Forest:
    """vgk_forest: the tail forests of a batch of tails, resident in HBM; `.graph` is the forest as one ResidentGraph-like handle
    whose windows are the trees (usable with Engine.pack_windows / align_windows)."""

    class _Graph:
        def __init__(self, h):
            self.h = h

    def __init__(self, eng, h):
        self.eng = eng; self.h = h
        lib = eng.lib
        lib.vgk_forest_size.restype = ctypes.c_uint64; lib.vgk_forest_size.argtypes = [ctypes.c_void_p]
        lib.vgk_forest_graph.restype = ctypes.c_void_p; lib.vgk_forest_graph.argtypes = [ctypes.c_void_p]
        lib.vgk_forest_fetch.argtypes = [ctypes.c_void_p] * 4
        lib.vgk_forest_destroy.argtypes = [ctypes.c_void_p]; lib.vgk_forest_destroy.restype = None
        self.size = int(lib.vgk_forest_size(h))
        g = lib.vgk_forest_graph(h)
        self.graph = Forest._Graph(ctypes.c_void_p(g)) if g else None
        eng._indexes.add(self)

    def fetch(self):
        """-> (parent, node, length) per tree node: parent = index in the forest or -1, node = oriented node of the index"""
        parent = np.zeros(max(self.size, 1), dtype=np.int32); node = np.zeros(max(self.size, 1), dtype=np.uint32); length = np.zeros(max(self.size, 1), dtype=np.uint32)
        self.eng._check(self.eng.lib.vgk_forest_fetch(self.h, parent.ctypes.data, node.ctypes.data, length.ctypes.data), "vgk_forest_fetch")
        return parent[:self.size], node[:self.size], length[:self.size]

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.eng, "h", None):
                self.eng.lib.vgk_forest_destroy(self.h)
            self.h = None; self.graph = None

    __del__ = close


class MinimizerIndex:
    """vgk_minimizer_index: the (k, w)-minimizers of the haplotype threads with their graph positions, resident in HBM."""

    def __init__(self, eng, nodes, threads, k=29, w=11):
        self.eng = eng; self.k, self.w = k, w
        if isinstance(nodes, tuple):
            self._len = np.ascontiguousarray(nodes[0], dtype=np.uint32); self._seq = np.ascontiguousarray(nodes[1], dtype=np.uint8); nodes = self._len
        else:
            self._len = np.array([len(s) for s in nodes], dtype=np.uint32)
            self._seq = np.frombuffer("".join(nodes).encode(), dtype=np.uint8).copy()
        self._toff = np.concatenate([[0], np.cumsum([len(t) for t in threads])]).astype(np.uint32)
        self._tn = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.uint32) for t in threads]) if len(threads) and self._toff[-1] else np.zeros(1, np.uint32), dtype=np.uint32)
        d = Haplotypes(len(nodes), self._len.ctypes.data, self._seq.ctypes.data, len(threads), self._toff.ctypes.data, self._tn.ctypes.data)
        h = ctypes.c_void_p()
        eng.lib.vgk_minimizer_index_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        eng._check(eng.lib.vgk_minimizer_index_create(eng.h, ctypes.byref(d), k, w, ctypes.byref(h)), "vgk_minimizer_index_create")
        self.h = h
        eng.lib.vgk_minimizer_index_keys.restype = ctypes.c_uint64; eng.lib.vgk_minimizer_index_keys.argtypes = [ctypes.c_void_p]
        self.keys = int(eng.lib.vgk_minimizer_index_keys(h))
        eng._indexes.add(self)

    def set_policy(self, hit_cap=10, hard_hit_cap=500, score_fraction=0.9, on=True, paired=False):
        """vgk_minimizer_set_policy: find_seeds' choice of minimizers (giraffe's short-read defaults) for the following minimizer_seeds calls;
        on=False: none"""
        class Policy(ctypes.Structure):
            _fields_ = [("hit_cap", ctypes.c_uint32), ("hard_hit_cap", ctypes.c_uint32), ("minimizer_score_fraction", ctypes.c_double), ("paired", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]
        self.eng.lib.vgk_minimizer_set_policy.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        p = Policy(hit_cap, hard_hit_cap, score_fraction, 1 if paired else 0, 0)
        self.eng._check(self.eng.lib.vgk_minimizer_set_policy(self.h, ctypes.byref(p) if on else None), "vgk_minimizer_set_policy")

    def fetch(self):
        """vgk_minimizer_index_fetch: every indexed occurrence as (key, oriented node, offset), sorted"""
        lib = self.eng.lib
        lib.vgk_minimizer_index_hits.restype = ctypes.c_uint64; lib.vgk_minimizer_index_hits.argtypes = [ctypes.c_void_p]
        n = int(lib.vgk_minimizer_index_hits(self.h))
        hits = np.zeros(max(n, 1), dtype=MINIMIZER_HIT_DT)
        lib.vgk_minimizer_index_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        self.eng._check(lib.vgk_minimizer_index_fetch(self.h, hits.ctypes.data, n), "vgk_minimizer_index_fetch")
        return hits[:n]

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.eng, "h", None):
                self.eng.lib.vgk_minimizer_index_destroy.argtypes = [ctypes.c_void_p]; self.eng.lib.vgk_minimizer_index_destroy.restype = None
                self.eng.lib.vgk_minimizer_index_destroy(self.h)
            self.h = None

    __del__ = close


class ResidentGraph:
    """vgk_dgraph: one topologically ordered DAG kept in HBM; problems name windows (runs of consecutive nodes) of it."""

    def __init__(self, eng, node_len, seq, pred_off, pred_idx):
        self.eng = eng
        self.node_len = np.ascontiguousarray(node_len, dtype=np.uint32)
        self.seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self.pred_off = np.ascontiguousarray(pred_off, dtype=np.uint32)
        self.pred_idx = np.ascontiguousarray(pred_idx if len(pred_idx) else np.zeros(1, np.uint32), dtype=np.uint32)
        self.col = np.concatenate([[0], np.cumsum(self.node_len, dtype=np.int64)])
        g = np.zeros(1, dtype=GRAPH_DT)
        g["n_nodes"] = len(self.node_len); g["node_len"] = self.node_len.ctypes.data; g["seq"] = self.seq.ctypes.data
        g["pred_off"] = self.pred_off.ctypes.data; g["pred_idx"] = self.pred_idx.ctypes.data
        h = ctypes.c_void_p()
        eng._check(eng.lib.vgk_graph_create(eng.h, g.ctypes.data, ctypes.byref(h)), "vgk_graph_create")
        self.h = h
        eng._indexes.add(self)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.eng, "h", None):
                self.eng.lib.vgk_graph_destroy(self.h)
            self.h = None

    __del__ = close


class WindowSet:
    """A batch of window problems (vgk_window_problem): reads in one flat ASCII array, read i = reads[read_off[i]:read_off[i+1]],
    aligned against nodes [first_node[i], first_node[i] + n_nodes[i]) of a ResidentGraph."""

    def __init__(self, reads, read_off, first_node, n_nodes, flags, max_gap=None, cols=None):
        self.reads = np.ascontiguousarray(reads, dtype=np.uint8)
        self.read_off = np.asarray(read_off, dtype=np.int64)
        n = self.n = len(self.read_off) - 1
        arr = np.zeros(n, dtype=WINDOW_DT)
        arr["read_off"] = self.read_off[:-1]; arr["read_len"] = np.diff(self.read_off)
        arr["flags"] = flags; arr["first_node"] = first_node; arr["n_nodes"] = n_nodes
        if max_gap is not None:
            arr["max_gap_length"] = np.asarray(max_gap, dtype=np.uint32)
        self.array = arr
        self.cols = None if cols is None else np.asarray(cols, dtype=np.int64)      # graph bases per window (sizes the default op array)

    @property
    def seq_off(self):
        c = self.cols if self.cols is not None else np.zeros(self.n, dtype=np.int64)
        return np.concatenate([[0], np.cumsum(c)])


class ExtensionSet:
    """A batch of extension windows (vgk_extension_problem): one pass of a seeded X-drop alignment from a position inside a window."""

    def __init__(self, reads, read_off, first_node, n_nodes, flags, max_gap, start_node, start_offset, query_offset, leftward, cols=None):
        self.reads = np.ascontiguousarray(reads, dtype=np.uint8)
        self.read_off = np.asarray(read_off, dtype=np.int64)
        n = self.n = len(self.read_off) - 1
        arr = np.zeros(n, dtype=EXTENSION_DT)
        arr["read_off"] = self.read_off[:-1]; arr["read_len"] = np.diff(self.read_off)
        for k, v in (("flags", flags), ("first_node", first_node), ("n_nodes", n_nodes), ("max_gap_length", max_gap), ("start_node", start_node),
                     ("start_offset", start_offset), ("query_offset", query_offset), ("leftward", leftward)):
            arr[k] = np.asarray(v, dtype=np.uint32)
        self.array = arr
        self.cols = None if cols is None else np.asarray(cols, dtype=np.int64)

    @property
    def seq_off(self):
        c = self.cols if self.cols is not None else np.zeros(self.n, dtype=np.int64)
        return np.concatenate([[0], np.cumsum(c)])


class GaplessSet:
    """A batch of gapless-extension problems (vgk_gapless_problem) over shared numpy arenas."""

    def __init__(self, reads, read_off, seeds, seed_off, max_mismatches=4, overlap_threshold=0.8, trim=True, node_cap=None, mism_cap=None):
        self.reads = np.ascontiguousarray(reads, dtype=np.uint8)
        self.read_off = np.asarray(read_off, dtype=np.int64)
        self.seeds = np.ascontiguousarray(seeds, dtype=SEED_DT)
        self.seed_off = np.asarray(seed_off, dtype=np.int64)
        n = self.n = len(self.read_off) - 1
        arr = np.zeros(n, dtype=GAPLESS_DT)
        arr["read"] = self.reads.ctypes.data + self.read_off[:-1]
        arr["read_len"] = np.diff(self.read_off)
        arr["n_seeds"] = np.diff(self.seed_off)
        arr["seeds"] = self.seeds.ctypes.data + 8 * self.seed_off[:-1]
        arr["max_mismatches"] = max_mismatches
        arr["flags"] = np.where(np.broadcast_to(np.asarray(trim), (n,)), VGK_GAPLESS_TRIM, 0)
        arr["overlap_threshold"] = overlap_threshold
        self.array = arr
        rl = np.diff(self.read_off); ns = np.diff(self.seed_off)
        self.ext_cap = int(ns.sum()) + 1
        self.node_cap = int(node_cap if node_cap is not None else (ns * (rl + 2)).sum()) + 1
        self.mism_cap = int(mism_cap if mism_cap is not None else (ns * rl).sum()) + 1

    @classmethod
    def from_lists(cls, problems):
        reads = [np.frombuffer(p["read"].encode(), dtype=np.uint8) for p in problems]
        read_off = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64)
        read_buf = np.concatenate(reads) if len(reads) and read_off[-1] else np.zeros(1, np.uint8)
        seed_off = np.concatenate([[0], np.cumsum([len(p["seeds"]) for p in problems])]).astype(np.int64)
        seeds = np.zeros(max(int(seed_off[-1]), 1), dtype=SEED_DT)
        k = 0
        for p in problems:
            for node, diff in p["seeds"]:
                seeds[k] = (node, diff); k += 1
        return cls(read_buf, read_off, seeds, seed_off, [p.get("max_mismatches", 4) for p in problems],
                   [p.get("overlap_threshold", 0.8) for p in problems], [p.get("trim", True) for p in problems])


class WfaSet:
    """A batch of WFA problems (vgk_wfa_problem) over one shared numpy arena of sequences."""

    def __init__(self, seqs, seq_off, mode, from_node, from_offset, to_node, to_offset, path_cap=None, edit_cap=None):
        self.seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        self.seq_off = np.asarray(seq_off, dtype=np.int64)
        n = self.n = len(self.seq_off) - 1
        arr = np.zeros(n, dtype=WFA_DT)
        arr["seq"] = self.seqs.ctypes.data + self.seq_off[:-1]
        arr["seq_len"] = np.diff(self.seq_off)
        arr["mode"] = mode
        arr["from_node"] = from_node; arr["from_offset"] = from_offset
        arr["to_node"] = to_node; arr["to_offset"] = to_offset
        self.array = arr
        sl = np.diff(self.seq_off)
        self.path_cap = int(path_cap if path_cap is not None else (4 * sl + 64).sum()) + 1
        self.edit_cap = int(edit_cap if edit_cap is not None else (2 * sl + 8).sum()) + 1

    @classmethod
    def from_lists(cls, problems):
        modes = {"connect": WFA_CONNECT, "suffix": WFA_SUFFIX, "prefix": WFA_PREFIX}
        seqs = [np.frombuffer(p["seq"].encode(), dtype=np.uint8) for p in problems]
        seq_off = np.concatenate([[0], np.cumsum([len(r) for r in seqs])]).astype(np.int64)
        buf = np.concatenate(seqs) if len(seqs) and seq_off[-1] else np.zeros(1, np.uint8)
        none = (WFA_NO_NODE, 0)
        return cls(buf, seq_off, [modes[p.get("mode", "connect")] for p in problems],
                   [(p.get("from") or none)[0] for p in problems], [(p.get("from") or none)[1] for p in problems],
                   [(p.get("to") or none)[0] for p in problems], [(p.get("to") or none)[1] for p in problems])


class HaploIndex:
    """The haplotype index the gapless extender walks (stands in for vg's GBWTGraph)."""

    def __init__(self, eng, nodes, threads=None, gbwt=None):
        """threads: [[oriented node]]; or gbwt: the bytes of a GBWT file (vgk_haplo_create_gbwt)"""
        self.eng = eng
        if isinstance(nodes, tuple):                                   # (node lengths, concatenated forward bases) as arrays: graphs with millions of nodes
            self.nodes = None
            self._len = np.ascontiguousarray(nodes[0], dtype=np.uint32); self._seq = np.ascontiguousarray(nodes[1], dtype=np.uint8)
            nodes = self._len
        else:
            self.nodes = list(nodes)
            self._len = np.array([len(s) for s in nodes], dtype=np.uint32)
            self._seq = np.frombuffer("".join(nodes).encode(), dtype=np.uint8).copy()
        if gbwt is not None:
            h = ctypes.c_void_p()
            eng.lib.vgk_haplo_create_gbwt.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            eng._check(eng.lib.vgk_haplo_create_gbwt(eng.h, bytes(gbwt), len(gbwt), len(nodes), self._len.ctypes.data, self._seq.ctypes.data, ctypes.byref(h)),
                       "vgk_haplo_create_gbwt")
            self.h = h
            eng._indexes.add(self)
            return
        self._toff = np.concatenate([[0], np.cumsum([len(t) for t in threads])]).astype(np.uint32)
        self._tn = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.uint32) for t in threads]) if len(threads) and self._toff[-1] else np.zeros(1, np.uint32), dtype=np.uint32)
        d = Haplotypes(len(nodes), self._len.ctypes.data, self._seq.ctypes.data, len(threads), self._toff.ctypes.data, self._tn.ctypes.data)
        h = ctypes.c_void_p()
        eng._check(eng.lib.vgk_haplo_create(eng.h, ctypes.byref(d), ctypes.byref(h)), "vgk_haplo_create")
        self.h = h
        eng._indexes.add(self)

    def run_nodes(self):
        """nodes of the index's merged-run form, which the WFA wavefront kernel walks (= the graph's nodes when nothing merged)"""
        self.eng.lib.vgk_haplo_run_nodes.restype = ctypes.c_uint64; self.eng.lib.vgk_haplo_run_nodes.argtypes = [ctypes.c_void_p]
        return int(self.eng.lib.vgk_haplo_run_nodes(self.h))

    def search_nodes(self):
        """nodes of the index the gapless search walks (unary runs merged at build; = the graph's nodes when nothing merged)"""
        self.eng.lib.vgk_haplo_search_nodes.restype = ctypes.c_uint64; self.eng.lib.vgk_haplo_search_nodes.argtypes = [ctypes.c_void_p]
        return int(self.eng.lib.vgk_haplo_search_nodes(self.h))

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.eng, "h", None):
                self.eng.lib.vgk_haplo_destroy(self.h)
            self.h = None

    __del__ = close


class Batch:
    def __init__(self, eng, h, ps, ops_per_problem):
        self.eng, self.h, self.ps, self.ops_per = eng, h, ps, ops_per_problem

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.free()

    def free(self):
        if self.h:
            self.eng.lib.vgk_batch_free(self.h); self.h = None

    def run(self):
        self.eng._check(self.eng.lib.vgk_gssw_run(self.h), "vgk_gssw_run")

    def sync(self):
        self.eng._check(self.eng.lib.vgk_batch_sync(self.h), "vgk_batch_sync")

    def kernel_ms(self, which=-1):
        return self.eng.lib.vgk_batch_kernel_ms(self.h, which)

    def cells(self):
        return self.eng.lib.vgk_batch_cells(self.h)

    def alg_bytes(self):
        return self.eng.lib.vgk_batch_alg_bytes(self.h)

    def device_bytes(self):
        return self.eng.lib.vgk_batch_device_bytes(self.h)

    def lane(self):
        """launch lane (stream) of this batch: consecutive batches of a context alternate between two"""
        self.eng.lib.vgk_batch_lane.argtypes = [ctypes.c_void_p]
        return self.eng.lib.vgk_batch_lane(self.h)

    def wave_steps(self):
        return self.eng.lib.vgk_batch_wave_steps(self.h)

    def speculated(self):
        """did the last run fill without traceback codes first (the speculative fill; the context's feedback decides per run)"""
        self.eng.lib.vgk_batch_speculated.argtypes = [ctypes.c_void_p]
        return bool(self.eng.lib.vgk_batch_speculated(self.h))

    def fetch(self, into=None):
        """-> (results, ops).  `into` = (results, ops) arrays of an earlier fetch of a batch of the same shape, written again
        (a streaming caller keeps its output buffers; fresh numpy arrays cost a page fault per 4 KB touched)."""
        n = self.ps.n
        if self.ops_per:
            cap = n * self.ops_per
        else:
            cap = int(np.diff(self.ps.read_off).sum() + np.diff(self.ps.seq_off).sum() + 2 * n)
        if into is not None:
            res, ops = into[0], (into[1].base if into[1].base is not None else into[1])
            assert len(res) == n and len(ops) >= max(cap, 1)
        else:
            res = np.zeros(n, dtype=RESULT_DT)
            ops = np.zeros(max(cap, 1), dtype=OP_DT)
        written = ctypes.c_size_t()
        self.eng._check(self.eng.lib.vgk_gssw_fetch(self.h, res.ctypes.data, ops.ctypes.data, cap, ctypes.byref(written)),
                        "vgk_gssw_fetch")
        return res, ops[:written.value]


def cigar_string(res_row, ops):
    """'offset@node:3M1D,node:...' for debugging / comparison."""
    o = ops[res_row["ops_begin"]:res_row["ops_begin"] + res_row["n_ops"]]
    parts, cur = [], None
    for e in o:
        if cur != e["node"]:
            parts.append("%d:" % e["node"]); cur = e["node"]
        parts[-1] += "%d%s" % (e["len"], OP_CHARS[e["op"]])
    return "%d@%s" % (res_row["first_offset"], ",".join(parts))
