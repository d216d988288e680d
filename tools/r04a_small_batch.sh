set -x
mkdir -p gpurun_out/r04a
cd $GRAFT_REPO_ROOT
for spec in "cfg3 8192 100" "cfg3 1024 50" "cfg5 4096 70" "cfg5 1024 50" "cfg3 4096 100" "cfg3 16384 100"; do
  set -- $spec
  python bench.py --workload $1 --scenarios $2 --periods $3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04a/${1}_${2}_${3}_eager.json 2> gpurun_out/r04a/${1}_${2}_${3}_eager.err
  python bench.py --workload $1 --scenarios $2 --periods $3 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r04a/${1}_${2}_${3}_eager_notimer.json 2>> gpurun_out/r04a/${1}_${2}_${3}_eager.err
  python bench.py --workload $1 --scenarios $2 --periods $3 --steps 5 --warmup 2 --no-cpu-baseline --graph > gpurun_out/r04a/${1}_${2}_${3}_graph.json 2> gpurun_out/r04a/${1}_${2}_${3}_graph.err
done
