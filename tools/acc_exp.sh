# Timing of k_exact_accumulate on the converged bench (GPU box); env assignments as arguments
cd $GRAFT_REPO_ROOT
run() { echo -n "[$*] "; env "$@" python bench.py --cpu-sample 0 --start planted --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['roofline']['kernels_ms'])"; }
run X=1
run SPKM_PTS=8
run SPKM_ACC_THREADS=512 SPKM_ACC_BLOCKS=2 SPKM_PTS=16
