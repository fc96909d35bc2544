#!/bin/bash
# Where a wavefront of k_icp_pass spends its time: builds icp.hip with -DO3D_PASS_PROF=1 into variants/libpprof.so (the 100 MHz
# clock at the phase boundaries, summed by lane 0 of every wavefront; first-in / last-out times; printed and cleared by the host
# after every pass), runs scripts/search_ab.py with it and restores the library.
# Build here (no GPU needed): scripts/pass_phases.sh build ; on the GPU box: scripts/pass_phases.sh run
set -e
if [ "$1" = build ]; then
  cd open3d_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function \
    -D__HIP_PLATFORM_AMD__ -DO3D_PASS_PROF=1 -c icp.hip -o /tmp/icp_pprof.o
  mkdir -p ../../variants
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../../variants/libpprof.so capi.o grid.o /tmp/icp_pprof.o sort_scan.o \
    preprocess.o ransac.o xfer.o synthetic.o
  cp ../libo3d_amd.so ../../variants/libcur.so
else
  cp variants/libpprof.so open3d_amd/libo3d_amd.so
  rm -f gpurun_out/pass_marks.bin
  O3D_PASS_PROF_DUMP=gpurun_out/pass_marks.bin AB_ONLY=p2l_r2 timeout -k 10 200 python3 scripts/search_ab.py pprof > gpurun_out/pprof.json 2> gpurun_out/pprof.err || true
  cp variants/libcur.so open3d_amd/libo3d_amd.so
  grep "pass milestones" gpurun_out/pprof.err | head -24
fi
