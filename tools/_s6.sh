B="--gpus 1 --steps 640 --warmup 32 --group 1 --no-extra-legs --no-config-legs --cpu-seconds 0 --prewarm 0.05"
for i in 1 2 3; do
python bench.py $B 2>/dev/null | tail -1 > gpurun_out/s6_g1_default_$i.json
python bench.py $B --pipeline library 2>gpurun_out/s6_lib_$i.err | tail -1 > gpurun_out/s6_g1_library_$i.json
done
for g in 2 4; do
python bench.py --gpus 1 --steps 640 --warmup 32 --group $g --no-extra-legs --no-config-legs --cpu-seconds 0 --prewarm 0.05 2>/dev/null | tail -1 > gpurun_out/s6_g${g}_default_1.json
python bench.py --gpus 1 --steps 640 --warmup 32 --group $g --no-extra-legs --no-config-legs --cpu-seconds 0 --prewarm 0.05 --pipeline library 2>/dev/null | tail -1 > gpurun_out/s6_g${g}_library_1.json
done
for f in gpurun_out/s6_g*.json; do python -c "
import json,sys
d=json.loads(open('$f').read())
print('$f', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])
"; done
