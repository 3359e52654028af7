for q in 8 16; do for d in 4 5 6 8 10; do echo -n "K=20 HWQ=$q depth=$d: "; GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --depth $d --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['chunks_per_s'], d['ms_per_step'])"; done; done
