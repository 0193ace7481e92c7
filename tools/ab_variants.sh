#!/bin/bash
# A/B of library variants (make variant VARIANT=x EXTRA=-D...) on the 298 k room from inside (tools/gpu_interior.py) and from outside (tools/gpu_scenes_time.py c2),
# two interleaved rounds.  usage: tools/ab_variants.sh base wl9 wl16 ...
cd /root/repo
for rnd in 1 2; do
  for v in "$@"; do
    lib=/root/repo/software-raytracing_amd/libraylib_$v.so; [ "$v" = base ] && lib=/root/repo/software-raytracing_amd/libraylib.so
    RAYLIB_QUIET=1 RAYLIB_LIB=$lib SPP=32 python tools/gpu_interior.py $v 2>&1 | grep "spp"
    RAYLIB_QUIET=1 RAYLIB_LIB=$lib python tools/gpu_scenes_time.py c2 2>&1 | grep triangles | sed "s/^/$v /" | cut -c1-110
  done
done
