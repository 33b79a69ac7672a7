cd $GRAFT_REPO_ROOT; O=gpurun_out/r04p; mkdir -p $O
timeout -s KILL 1500 python tools/band_vs_exact_config2.py --batches 6 --reads 1000000 > $O/band_vs_exact_config2.json 2> $O/band_vs_exact_config2.err; echo "rc=$?"
tail -8 $O/band_vs_exact_config2.err; cat $O/band_vs_exact_config2.json
