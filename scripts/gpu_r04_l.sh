# round 4, call L: stamps of the two-cell one-launch decoders (diagnostics build)
set -x
LAS_CXXFLAGS=-DLAS_STAMPS LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python phones-las_amd/build.py --force 2>&1 | tail -2
for cfg in default-arch two-cell-bottom-only; do
CFG=$cfg LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python scripts/gpu_dec2_stamps.py 2>&1 | tail -45
done
