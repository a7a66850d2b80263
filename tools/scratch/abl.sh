for v in "" abl1 abl2 abl4 abl8 abl3 abl6; do
  lib=""; [ -n "$v" ] && lib=tools/scratch/libfvp_hip_$v.so
  FVP_LIB=$lib FVP_BB_DMA_RING=${RING:-1} timeout 200 python tools/bench_backbone.py --images 40 --per-op 2>&1 | python3 -c "
import sys,re
o={}
for l in sys.stdin:
    m=re.search(r'op\s*(\d+) kind.*?([\d.]+) us', l)
    if m: o[int(m[1])]=float(m[2])
print('${v:-base}', ' '.join(f'{k}:{o[k]:.0f}' for k in (14,17,26,28,29,30,45,46,47,48,49,54,55,56) if k in o))
"
done
