# A/B of the screen's work-saving layers on the bench workload; usage: bash tools/hint_sweep.sh [bench args]
run() {
  echo -n "$*: "; env "$@" python bench.py --cpu-sample 0 $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; print(round(d['value'],2), 'it/s', round(d['ms_per_step'],2), 'ms', {k: round(v,2) for k,v in r['kernels_ms'].items()}, c['screen_rounds_last_iter'], c['screen_form_last_iter'], 'early', c['early_finished_steps'], 'skipped', c['skipped_steps_last_iter'], 'listed', c['uncertified_points_last_iter'], 'share', round(r['screen_steps_processed_share'],3))"
}
ARGS="$*"
run X=1
run SPKM_NO_BOUNDS=1
run SPKM_NO_BOUNDS=1 SPKM_NO_HINT=1
run SPKM_NO_BOUNDS=1 SPKM_NO_HINT=1 SPKM_NO_PRUNE=1
run SPKM_NO_SCREEN=1
