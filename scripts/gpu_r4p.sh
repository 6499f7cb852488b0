# Round 4, GPU call p: (i) does kernarg preloading (-mllvm -amdgpu-kernarg-preload-count) take anything off a dependent
# launch on this runtime?  scripts/micro/launch_floor built with and without it; (ii) navigation with 8 waves per tile (the
# 16-wave kernels of the 8 192-environment shard sit at the 128-register cap with scratch; 65 536 runs 4 waves per tile)
TAG=r04p
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
{ echo "# without preload"; $S/micro/launch_floor 32768 8 | grep -E "K0 empty  |K1|K2 tile copy|K3"; echo "# with -mllvm -amdgpu-kernarg-preload-count=12"; $S/micro/launch_floor_preload 32768 8 | grep -E "K0 empty  |K1|K2 tile copy|K3"; echo "# without, again"; $S/micro/launch_floor 32768 8 | grep -E "K0 empty  |K1|K2 tile copy|K3"; } > $OUT/${TAG}_kernarg_preload_launch_floor.txt 2>&1; cat $OUT/${TAG}_kernarg_preload_launch_floor.txt
{ for L in 0 8 4; do for B in 8192 65536; do
    if [ $L = 0 ]; then ACTIONS=zero python $S/bench_bound.py navigation $B; else LANES=$L ACTIONS=zero python $S/bench_bound.py navigation $B; fi
  done; done; } 2>&1 | grep "^{" > $OUT/${TAG}_navigation_waves_per_tile.jsonl; cut -c1-300 $OUT/${TAG}_navigation_waves_per_tile.jsonl
