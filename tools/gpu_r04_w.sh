cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04w; mkdir -p $O
timeout -s KILL 120 ./build/pinned_write > $O/pinned_write.txt 2>&1 < /dev/null; echo "rc=$?"; cat $O/pinned_write.txt
