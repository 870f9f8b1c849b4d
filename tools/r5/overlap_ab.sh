set -x
python -m pytest tests/test_clip_model_gpu.py -x -q -m gpu 2>&1 | tail -5
for v in 0 1 0 1; do
UNIIR_OVERLAP_TOWERS=$v python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-retrieval --no-unpacked 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('OVERLAP=$v', d['value'], d['ms_per_step'])
"
done
