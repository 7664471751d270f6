#!/bin/bash
# rebuild ONLY lde_pendulum.o (+ link) after a change to lde_pend_lp.h / lde_pendulum.hip: build.py would recompile every source (a header changed)
set -e
cd "$(dirname "$0")/../latentdiffeq.jl_amd"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $1 -c csrc/lde_pendulum.hip -o ${2:-_obj}/lde_pendulum.o
if [ -z "$2" ]; then
  touch _obj/*.o _obj/lde_mlp.o.checked
  hipcc --offload-arch=gfx950 -shared -fPIC -o liblde.so _obj/lde_api.o _obj/lde_pendulum.o _obj/lde_mlp.o _obj/lde_chain.o _obj/lde_rnn.o _obj/lde_loss.o _obj/lde_optim.o _obj/lde_comm.o _obj/lde_buildinfo.o -ldl
fi
