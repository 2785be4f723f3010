# A/B of the hinted two-phase screen's knobs on the bench workload (SPKM_HINT_A rounds for all centroids,
# SPKM_HINT_C squared distance factor); usage: bash tools/hint_sweep.sh "A=.. C=.." ...
run() {
  echo -n "$*: "; env "$@" python bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['kernel_ms'],2), c['screen_rounds_last_iter'], c['screen_form_last_iter'], c['early_finished_steps'], c['uncertified_points_last_iter'])"
}
run X=1
run SPKM_NO_HINT=1
run SPKM_QUAD_EQUAL=1
run SPKM_SHARE_EXTRA=1
run SPKM_NO_FUSE=1
run SPKM_QUAD_W=0.8
run SPKM_QUAD_W=1.0
run SPKM_QUAD_W=1.2
run SPKM_CHUNK=4096
run SPKM_CHUNK=16384
run SPKM_CHUNK=65536
