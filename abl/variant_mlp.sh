#!/bin/bash
# build a variant of liblde.so that differs in lde_mlp.o only:  abl/variant_mlp.sh <tag> "<extra hipcc flags>"  →  abl/liblde_<tag>.so
set -e
cd "$(dirname "$0")/../latentdiffeq.jl_amd"
tag=$1; shift
mkdir -p /tmp/lde_var_$tag
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $@ -c csrc/lde_mlp.hip -o /tmp/lde_var_$tag/lde_mlp.o
python3 check_agprs.py /tmp/lde_var_$tag/lde_mlp.o > /tmp/lde_var_$tag/agpr.log 2>&1 || { echo "AGPR check failed for $tag"; tail -3 /tmp/lde_var_$tag/agpr.log; exit 1; }
hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/liblde_$tag.so _obj/lde_api.o _obj/lde_pendulum.o /tmp/lde_var_$tag/lde_mlp.o _obj/lde_chain.o _obj/lde_rnn.o _obj/lde_loss.o _obj/lde_optim.o _obj/lde_comm.o _obj/lde_buildinfo.o -ldl
echo built abl/liblde_$tag.so
