#!/bin/bash
# Profiles of the round for profiles/ (run on the GPU box; summaries: python tools/summarize_profiles.py gpurun_out/prof_round r04):
#   headline bench: three PMC passes (SQ/GRBM, FETCH_SIZE, WRITE_SIZE + LDS conflicts), then the kernel stats pass
#   depth-4 bench and train mode: kernel stats + the same PMC passes
# Counters are collected in their own runs (--pmc with --kernel-trace only); every program after `--` is python3 itself.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_round
rm -rf $out && mkdir -p $out
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
PMC2="FETCH_SIZE"
PMC3="WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
TAG=${TAG:-r06}
run() {   # run <name> <bench args...>: the three PMC passes FIRST; for the headline run the traffic figure they give is written
          # to profiles/ (on this box) before the stats pass, so that the bench line of the stats pass -- the line that gets
          # committed -- quotes the traffic measured beside it (tools/summarize_profiles.py --traffic-only)
  name=$1; shift
  mkdir -p $out/${name}_stats $out/${name}_pmc1 $out/${name}_pmc2 $out/${name}_pmc3
  rocprofv3 --kernel-trace --pmc $PMC1 --output-format csv -d $out/${name}_pmc1 -- python3 bench.py "$@" > $out/${name}_pmc1/bench.log 2>&1
  rocprofv3 --kernel-trace --pmc $PMC2 --output-format csv -d $out/${name}_pmc2 -- python3 bench.py "$@" > $out/${name}_pmc2/bench.log 2>&1
  rocprofv3 --kernel-trace --pmc $PMC3 --output-format csv -d $out/${name}_pmc3 -- python3 bench.py "$@" > $out/${name}_pmc3/bench.log 2>&1
  echo "$name pmc done"
  if [ $name = n100k ]; then python3 tools/summarize_profiles.py $out $TAG --traffic-only && cp profiles/${TAG}_update_kernel_traffic.json $out/; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${name}_stats -- python3 bench.py "$@" > $out/${name}_stats/bench.log 2>&1
  echo "$name stats done"
}
run n100k --steps 2 --warmup 2 --no-cpu-baseline --no-other-configs
run depth4 --config dsmgp_n100k_d8_depth4 --steps 2 --warmup 2 --no-cpu-baseline
run train --mode train --steps 2 --warmup 1
# BASELINE configs 2 and 3 (tools/run_config3.py: single GP n = 4096, then the PoE of 128 ArdSE experts): kernel stats, the SQ / GRBM
# pass, and the vector-instruction counters that say whether the ArdSE kernel function (8 exp per entry) is what its launches do
mkdir -p $out/c23_stats $out/c23_pmc1 $out/c23_pmc4
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c23_stats -- python3 tools/run_config3.py > $out/c23_stats/run.log 2>&1
rocprofv3 --kernel-trace --pmc $PMC1 --output-format csv -d $out/c23_pmc1 -- python3 tools/run_config3.py > $out/c23_pmc1/run.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d $out/c23_pmc4 -- python3 tools/run_config3.py > $out/c23_pmc4/run.log 2>&1 || echo "c23 pmc4 pass failed (counter names?)"
echo "c23 done"
python3 tools/fit_timeline.py $out/depth4_stats $TAG > $out/fit_timeline.log 2>&1 || echo "fit_timeline failed"
echo profiled
