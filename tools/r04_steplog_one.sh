set -e
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04m; mkdir -p $o
DSMGP_STEPLOG=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-launch > $o/h_ahead.json 2> $o/h_ahead.txt
DSMGP_STEPLOG=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $o/h_one.json 2> $o/h_one.txt
DSMGP_STEPLOG=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-launch --simulate-shard 0/8 > $o/s_ahead.json 2> $o/s_ahead.txt
DSMGP_STEPLOG=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --simulate-shard 0/8 > $o/s_one.json 2> $o/s_one.txt
echo ok
