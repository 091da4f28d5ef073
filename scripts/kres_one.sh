#!/bin/bash
# VGPR / occupancy of the two tiled kernels only (fast: one small TU). usage: kres_one.sh [extra -D flags]
mkdir -p build/prof && cd build/prof
cat > one.hip <<'EOS'
#include "../../fastdem_amd/csrc/fdm_tiled.hpp"
using namespace fdm;
template __global__ void fdm::k_tbin<true,false,256,true>(const ScanParams, const GeomConst, const TileGrid, DevState*, const ScanInputs, const Scratch, const TilePool, int32_t*);
template __global__ void fdm::k_tupdate<KalmanRecPolicy, true, false>(const ScanParams, const GeomConst, const TileGrid, DevState*, const KalmanRecLayers, float* const*, int, const TilePool, const TileAux, unsigned);
template __global__ void fdm::k_tupdate<P2RecPolicy, false, true>(const ScanParams, const GeomConst, const TileGrid, DevState*, const P2RecLayers, float* const*, int, const TilePool, const TileAux, unsigned);
EOS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math "$@" -Rpass-analysis=kernel-resource-usage -c one.hip -o one.o 2>&1 | awk '
/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
/ VGPRs:/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v)}
/Occupancy \[waves/ {oc=$0; sub(/.*: /,"",oc); sub(/ .*/,"",oc)}
/LDS Size/ {print v, oc, name}' | while read v oc name; do echo "$v $oc $(echo $name | c++filt | cut -c1-60)"; done
