#!/bin/bash
# Wave cycles per phase of a search trip (k_icp_search_coop / the search half of k_icp_pass): builds icp.hip with
# -DO3D_SRCH_PROF=1 into variants/libsprof.so (s_memtime at the phase boundaries, summed by lane 0 of every wavefront, printed
# and cleared by the host after every pass), runs scripts/search_ab.py with it and restores the library.
# Build here (no GPU needed): scripts/search_phases.sh build ; on the GPU box: scripts/search_phases.sh run
set -e
if [ "$1" = build ]; then
  cd open3d_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function \
    -D__HIP_PLATFORM_AMD__ -DO3D_SRCH_PROF=1 -c icp.hip -o /tmp/icp_prof.o
  mkdir -p ../../variants
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../../variants/libsprof.so capi.o grid.o /tmp/icp_prof.o sort_scan.o \
    preprocess.o ransac.o xfer.o synthetic.o
  cp ../libo3d_amd.so ../../variants/libcur.so
else
  cp variants/libsprof.so open3d_amd/libo3d_amd.so
  timeout -k 10 200 python3 scripts/search_ab.py sprof > gpurun_out/sprof.json 2> gpurun_out/sprof.err || true
  cp variants/libcur.so open3d_amd/libo3d_amd.so
  grep "search phases" gpurun_out/sprof.err | head -30
fi
