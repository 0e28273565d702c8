# In-kernel phase stamps of the tiled MFMA convolution (wave 0 of workgroup 0), per work item, and the clock the chip holds;
# needs the tuning build   python -m mri_epilepsy_diagnosis_amd.build --variant stamps -DMRI3D_EXPERIMENT_STAMPS
# reps are sized for ~2 s of back-to-back launches per line (DVFS settles).
lib="--lib mri_epilepsy_diagnosis_amd/libmri3d_hip_stamps.so"
python tools/conv_bench.py $lib 8 16 160 192 160 2 2500 fwd 2>/dev/null
python tools/conv_bench.py $lib 16 16 160 192 160 2 1500 fwd,dgrad 2>/dev/null
python tools/conv_bench.py $lib 32 32 80 96 80 2 3000 fwd 2>/dev/null
python tools/conv_bench.py $lib --cat 16 48 16 160 192 160 2 700 fwd,dgrad 2>/dev/null
python tools/conv_bench.py $lib --cat 32 96 32 80 96 80 2 1200 fwd 2>/dev/null
python tools/conv_bench.py $lib 16 16 160 192 160 2 8000 fwd bf16 2>/dev/null
python tools/conv_bench.py $lib --cat 16 48 16 160 192 160 2 3000 fwd bf16 2>/dev/null
