#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats of the default bench command, then three SEPARATE counter passes (MI355X_MICROARCH.md "HBM" / "PMC slots":
#   FETCH_SIZE and WRITE_SIZE do not fit one pass; counters never together with --sys-trace)
# Output under gpurun_out/<tag>_*; tools/pmc_summary.py and tools/pmc_mfma_summary.py turn them into the tables under profiles/.
TAG=${1:-r06}
ONLY_STATS=${2:-0}      # 1: the kernel-trace pass and its timelines only
export TMPDIR=/tmp; cd "$(cd "$(dirname "$0")/.." && pwd)"
B="python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --pcie-steps 0 --jobs-in-flight 1 --no-side-legs"      # (the side legs would put extra passes behind the timed steps: the timelines below count steps from the end of the trace)
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -o ${TAG} --output-format csv -- $B > gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
if [ "$ONLY_STATS" != "1" ]; then
P="python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-roofline --pcie-steps 0 --jobs-in-flight 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/${TAG}_pmc_fetch -o f --output-format csv -- $P > gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/${TAG}_pmc_write -o w --output-format csv -- $P > gpurun_out/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/${TAG}_pmc_mfma -o m --output-format csv -- $P > gpurun_out/${TAG}_pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -d gpurun_out/${TAG}_pmc_issue -o i --output-format csv -- $P > gpurun_out/${TAG}_pmc_issue.log 2>&1
python3 tools/pmc_issue_summary.py gpurun_out/${TAG}_pmc_issue/i_counter_collection.csv gpurun_out/${TAG}_pmc_issue/i_kernel_trace.csv gpurun_out/${TAG}_pmc_issue_C3.csv > gpurun_out/${TAG}_pmc_issue.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_fetch/f_counter_collection.csv gpurun_out/${TAG}_pmc_write/w_counter_collection.csv gpurun_out/${TAG}_pmc_hbm_traffic_C3.csv > gpurun_out/${TAG}_pmc_hbm.txt 2>&1
python3 tools/pmc_mfma_summary.py gpurun_out/${TAG}_pmc_mfma/m_counter_collection.csv gpurun_out/${TAG}_pmc_mfma_C3.csv > gpurun_out/${TAG}_pmc_mfma.txt 2>&1
fi
# timelines out of the kernel trace of the first pass: idle gaps of the last step, one LM trial launch by launch, the analysis window
T=$(find gpurun_out/${TAG}_stats -name "${TAG}_kernel_trace.csv" | head -1)
STEP_FROM_END=2 python3 tools/trace_gaps.py $T 30 > gpurun_out/${TAG}_step_idle_gaps.txt 2>&1
TRIAL_FROM_END=8 python3 tools/trace_trial.py $T --all > gpurun_out/${TAG}_one_LM_trial_timeline.txt 2>&1
STEP_FROM_END=2 python3 tools/trace_window.py $T lc_kernel pg_front_asm_kernel > gpurun_out/${TAG}_step_analysis_window.txt 2>&1
STEP_FROM_END=2 python3 tools/trace_extract.py $T > gpurun_out/${TAG}_extraction_timeline.txt 2>&1
# keep what travels back small (gpurun merges at most 64 MiB): traces and raw counter dumps stay on the box, the reduced tables travel
rm -f gpurun_out/${TAG}_pmc_*/?_kernel_trace.csv gpurun_out/${TAG}_pmc_*/?_counter_collection.csv gpurun_out/${TAG}_stats/${TAG}_kernel_trace.csv
ls -la gpurun_out/${TAG}_* | head -40
