#!/bin/bash
# build a variant of liblde.so that differs in lde_pendulum.o only:  abl/variant_lib.sh <tag> "<extra hipcc flags>"  →  abl/liblde_<tag>.so
# (run the bench against it with LDE_LIB_PATH=abl/liblde_<tag>.so; abl/*.so are git-ignored)
set -e
cd "$(dirname "$0")/../latentdiffeq.jl_amd"
tag=$1; shift
mkdir -p /tmp/lde_var_$tag
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $@ -c csrc/lde_pendulum.hip -o /tmp/lde_var_$tag/lde_pendulum.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../abl/liblde_$tag.so _obj/lde_api.o /tmp/lde_var_$tag/lde_pendulum.o _obj/lde_mlp.o _obj/lde_chain.o _obj/lde_rnn.o _obj/lde_loss.o _obj/lde_optim.o _obj/lde_comm.o _obj/lde_buildinfo.o -ldl
echo built abl/liblde_$tag.so
