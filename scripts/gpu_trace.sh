# per-wave phase timeline: needs the -DVMAS_TRACE build of the library (rebuilt normally afterwards)
VMAS_HIPCC_EXTRA=-DVMAS_TRACE bash vectorizedmultiagentsimulator_amd/csrc/build.sh > /dev/null 2>&1
python scripts/trace_phases.py ${1:-8} ${2:-32768} 2>&1 | grep -v amdgpu | tail -${3:-16}
bash vectorizedmultiagentsimulator_amd/csrc/build.sh > /dev/null 2>&1
