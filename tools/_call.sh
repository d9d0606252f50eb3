out=gpurun_out/r06s; mkdir -p $out; rm -f $out/cfg4b.jsonl
M="python tools/model_bench.py --bf16 --act-bf16"
$M >> $out/cfg4b.jsonl 2>> $out/err.log
$M --checkpoint >> $out/cfg4b.jsonl 2>> $out/err.log
$M --checkpoint-levels 2 >> $out/cfg4b.jsonl 2>> $out/err.log
python tools/predict_bench.py 2>/dev/null | tail -2 | cut -c1-300
python tools/predict_bench.py --bf16 2>/dev/null | tail -1 | cut -c1-300
python - <<'PY'
import json
for ln in open('gpurun_out/r06s/cfg4b.jsonl'):
    if ln.startswith('{'):
        r=json.loads(ln); print(r.get('checkpoint_encoders'), r.get('ms_per_step'), {k:v for k,v in r.items() if 'mem' in k or 'gib' in k.lower()})
PY
