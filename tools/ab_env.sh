#!/bin/bash
# A/B of library variants and scene-build / render switches on the 298 k room from inside (tools/gpu_interior.py) and from outside (tools/gpu_scenes_time.py c2),
# interleaved rounds.  usage: [ROUNDS=2] [SPP=32] tools/ab_env.sh name[@lib][:ENV=VAL[,ENV=VAL...]] ...      (lib: libraylib_<lib>.so; default the tree's build)
cd /root/repo
for rnd in $(seq 1 ${ROUNDS:-2}); do
  for spec in "$@"; do
    name=${spec%%:*}; envs=""; [ "$spec" != "$name" ] && envs=${spec#*:}
    lib=/root/repo/software-raytracing_amd/libraylib.so
    tag=$name
    if [[ "$name" == *@* ]]; then lib=/root/repo/software-raytracing_amd/libraylib_${name#*@}.so; tag=${name%%@*}; fi
    (
      export RAYLIB_QUIET=1 RAYLIB_LIB=$lib
      IFS=',' read -ra KV <<< "$envs"; for kv in "${KV[@]}"; do [ -n "$kv" ] && export "$kv"; done
      SPP=${SPP:-32} python tools/gpu_interior.py $tag 2>&1 | grep "spp"
      [ -z "$NO_EXTERIOR" ] && python tools/gpu_scenes_time.py c2 2>&1 | grep triangles | sed "s/^/$tag /" | cut -c1-150
    )
  done
done
