# round 4, call W: phase stamps of the metric-M one-launch decoders (diagnostics build)
set -x
LAS_CXXFLAGS=-DLAS_STAMPS LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python phones-las_amd/build.py --force 2>&1 | tail -1
CFG=metric-M LAS_HIP_LIB=$PWD/phones-las_amd/liblas_hip_stamps.so python scripts/gpu_dec_stamps.py 2>&1 | tail -32
