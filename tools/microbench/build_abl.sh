#!/bin/bash
# ablation builds of ONE source file: tools/microbench/build_abl.sh conv_h16 ABL_NOMFMA ABL_NOEPI ...  -> exp/lib_<macro>.so
# (each macro compiles one part of the kernel out; results are garbage, the timing difference is that part's exposed cost)
# The #ifdef ABL_* switches are added to the kernel for the session and removed again before committing -- the tree carries
# none; DESIGN.md section 8 records what the round-1 ablations measured.
src=$1; shift
cd /root/repo
python ssl_cr_histo_amd/build.py > /dev/null
others=$(ls ssl_cr_histo_amd/build/*.o | grep -v "/$src.o")
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -D$m -x hip -c ssl_cr_histo_amd/csrc/$src.hip -o /tmp/abl_$m.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/lib_$m.so /tmp/abl_$m.o $others -L/opt/rocm/lib -lrccl && echo built exp/lib_$m.so
done
