mkdir -p gpurun_out/tl
for m in 31 1; do
cp profiles/tmp_libs/stamps_$m.so root_digger_amd/lib/librdamd.so
for c in 16 17 64; do
  timeout 600 python profiles/step_timeline.py --config c2 --repeat-classes $c --label "mask$m classes$c" 2>&1 | tee -a gpurun_out/tl/timeline.txt
done; done
