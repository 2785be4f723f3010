#!/bin/bash
# Host-side view of a short bench run (GPU box): rocprofv3 --hip-trace, then every HIP API call over 20 ms and every
# allocation / free / synchronisation from the first k_finalize_centers on, with arguments -- what the host pays between
# kernels (a 51-GB hipMalloc is ~1 s on a cold box; a hipFree of 400 MB waits for the device).
#   tools/hiptrace.sh [bench.py arguments; default --steps 4 --warmup 2 --cpu-sample 0 --no-regimes]
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for run in 1; do
rm -rf /tmp/ht1
rocprofv3 --hip-trace --kernel-trace --output-format json -d /tmp/ht1 -o t -- python $root/bench.py ${@:---steps 4 --warmup 2 --cpu-sample 0 --no-regimes} > /tmp/ht1.log 2>&1
tail -1 /tmp/ht1.log | cut -c1-200
f=$(find /tmp/ht1 -name "*.json" | head -1)
ls -la $f
python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d['rocprofiler-sdk-tool'][0]
api=r['buffer_records'].get('hip_api', [])
ops={}
for idx,k in enumerate(r['strings']['buffer_records']):
    ops[k.get('kind',idx)]=k['operations']
    ops[idx]=k['operations']
t0=min(a['start_timestamp'] for a in api)
kd=r['buffer_records'].get('kernel_dispatch', [])
ks={k['kernel_id']:k.get('formatted_kernel_name',k.get('kernel_name','?')) for k in r['kernel_symbols']}
fin=[(k['start_timestamp']-t0)/1e9 for k in kd if 'k_finalize_centers' in ks.get(k['dispatch_info']['kernel_id'],'')]
print('k_finalize_centers at', [round(v,3) for v in fin])
def nm(a):
    return (ops.get(a['kind']) or {})[a['operation']] if a['kind'] in ops else '?'
slow=[a for a in api if a['end_timestamp']-a['start_timestamp']>20e6 or (nm(a) in ('hipMalloc','hipFree','hipFuncSetAttribute','hipHostMalloc','hipStreamSynchronize','hipDeviceSynchronize') and (a['start_timestamp']-t0)/1e9 > fin[0]-0.06)]
for a in sorted(slow,key=lambda a:a['start_timestamp']):
    name=(ops.get(a['kind']) or {})[a['operation']] if a['kind'] in ops else f"kind{a['kind']}op{a['operation']}"
    print(f"{(a['start_timestamp']-t0)/1e9:8.3f} s {(a['end_timestamp']-a['start_timestamp'])/1e6:9.1f} ms {name} ", [(x['name'],x['value']) for x in a['args']][:4])
PY
done
