#!/bin/bash
# Profiles of the default bench command for profiles/: kernel stats + PMC passes (run on the GPU box), plus kernel stats
# of the depth-4 config and of the train mode.  Summaries: python tools/summarize_profiles.py gpurun_out/prof_round r02
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_round
rm -rf $out && mkdir -p $out/stats $out/pmc1 $out/pmc2 $out/pmc3 $out/stats_d4 $out/stats_train
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline > $out/stats/bench.log 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc1 -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline > $out/pmc1/bench.log 2>&1
echo pmc1 done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc2 -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline > $out/pmc2/bench.log 2>&1
echo pmc2 done
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc3 -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline > $out/pmc3/bench.log 2>&1
echo pmc3 done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_d4 -- python3 bench.py --config dsmgp_n100k_d8_depth4 --steps 2 --warmup 2 --no-cpu-baseline > $out/stats_d4/bench.log 2>&1
echo d4 done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_train -- python3 bench.py --mode train --steps 2 --warmup 1 > $out/stats_train/bench.log 2>&1
echo profiled
