export ACTIONS=zero
b() { python scripts/bench_bound.py navigation 16384 | tail -1 | cut -c130-260; }
echo -n "fresh box: "; b
python -c "import torch; x = torch.zeros(1, device='cuda'); torch.cuda.synchronize()"
echo -n "after a trivial process: "; b
echo -n "AMD_SERIALIZE_KERNEL=3: "; AMD_SERIALIZE_KERNEL=3 python scripts/bench_bound.py navigation 16384 | tail -1 | cut -c130-260
echo -n "again plain: "; b
echo -n "physics only (bench_world nav 16384 q1): "; QUEUES=1 python scripts/bench_world.py navigation 16384 500 | tail -1 | cut -c150-300
echo -n "HSA_ENABLE_SDMA=0: "; HSA_ENABLE_SDMA=0 python scripts/bench_bound.py navigation 16384 | tail -1 | cut -c130-260
echo -n "GPU_MAX_HW_QUEUES=2: "; GPU_MAX_HW_QUEUES=2 python scripts/bench_bound.py navigation 16384 | tail -1 | cut -c130-260
