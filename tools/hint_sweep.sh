# A/B of the hinted two-phase screen's knobs on the bench workload (SPKM_HINT_C squared distance factor); usage: bash tools/hint_sweep.sh "A=.. C=.." ...
run() {
  echo -n "$*: "; env "$@" python bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['kernel_ms'],2), c['screen_rounds_last_iter'], c['screen_form_last_iter'], c['early_finished_steps'], c['uncertified_points_last_iter'])"
}
run X=1
run SPKM_HINT_C=1.5
run SPKM_HINT_C=2
run SPKM_HINT_C=3
