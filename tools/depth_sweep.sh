for q in 8 16; do for d in 5 6 7 8; do echo -n "HWQ=$q depth=$d: "; GPU_MAX_HW_QUEUES=$q python bench.py --steps 40 --warmup 8 --depth $d --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['chunks_per_s'], d['ms_per_step'])"; done; done
echo -n "steps=20 depth=6: "; python bench.py --steps 20 --warmup 5 --depth 6 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['chunks_per_s'], d['ms_per_step'])"
echo -n "steps=20 depth=4: "; python bench.py --steps 20 --warmup 5 --depth 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['chunks_per_s'], d['ms_per_step'])"
