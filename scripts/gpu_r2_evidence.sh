# Round-2 baseline evidence (before any kernel change this round): kernel trace + PMC passes for the physics kernel of
# football 131072 / navigation 65536 / balance 1 M / balance 32768 and for the LIDAR kernel.
S=/root/repo/scripts
bash scripts/gpu_counters.sh ${PFX:-r02a}_football131072_physics 948 11900 131072 -- python $S/bench_world.py football 131072 200
bash scripts/gpu_counters.sh ${PFX:-r02a}_navigation65536_physics 672 1800 65536 -- python $S/bench_world.py navigation 65536 200
bash scripts/gpu_counters.sh ${PFX:-r02a}_balance1048576_physics 384 1700 1048576 -- python $S/bench_world.py balance 1048576 100
bash scripts/gpu_counters.sh ${PFX:-r02a}_balance32768_physics 384 1700 32768 -- python $S/bench_world.py balance 32768 300
bash scripts/gpu_counters.sh ${PFX:-r02a}_lidar65536 480 27000 65536 -- python $S/bench_lidar.py 65536
