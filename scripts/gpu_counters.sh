# rocprofv3 evidence for one command: kernel trace (durations) + PMC passes, each in its own run.
#   [RATED=kernel-name-part] bash scripts/gpu_counters.sh TAG BYTES_PER_ENV FLOP_PER_ENV ENVS -- CMD...
# writes gpurun_out/$EVIDENCE_DIR/TAG_kernel_stats.csv and TAG_pmc_summary.txt (copy to profiles/ to keep).  The bytes / flop
# per environment rate ONE kernel family (RATED = family[:physics|:env], default step_kernel_spec: scripts/pmc_summary.py); the others are listed without a roofline line.
TAG=$1; BPE=$2; FPE=$3; ENVS=$4; shift 5
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${EVIDENCE_DIR:-r04}; mkdir -p $OUT
W=/tmp/cnt_$TAG; rm -rf $W; mkdir -p $W
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace -o t -- "$@" > $W/trace.log 2>&1
f=$(find $W/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -v "at::\|rocclr\|Cijk" "$f" | head -12 > $OUT/${TAG}_kernel_stats.csv
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $W/p1 -o p -- "$@" > $W/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $W/p2 -o p -- "$@" > $W/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/p3 -o p -- "$@" > $W/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/p4 -o p -- "$@" > $W/p4.log 2>&1
cd $R
python scripts/pmc_summary.py $W $OUT/${TAG}_kernel_stats.csv $BPE $FPE $ENVS "$*" ${RATED:-step_kernel_spec} > $OUT/${TAG}_pmc_summary.txt 2>&1
cat $OUT/${TAG}_pmc_summary.txt
