#!/bin/bash
# Fast look at the large-scan kernels alone: one small TU with the configs[3] instantiations, resource usage and the
# .s with line tables (build/prof/one-hip-amdgcn-amd-amdhsa-gfx950.s).  usage: tiled_asm.sh [extra hipcc flags]
mkdir -p build/prof && cd build/prof
cat > one.hip <<'EOS'
#include "../../fastdem_amd/csrc/fdm_tiled.hpp"
using namespace fdm;
template __global__ void fdm::k_tbin<true,false,256,true>(const ScanParams, const GeomConst, const TileGrid, DevState*, const ScanInputs, const Scratch, const TilePool, int32_t*);
template __global__ void fdm::k_tupdate<KalmanRecPolicy, true, false>(const ScanParams, const GeomConst, const TileGrid, DevState*, const KalmanRecLayers, float* const*, int, const TilePool, const TileAux, const TileWork);
template __global__ void fdm::k_tupdate_tbin<KalmanRecPolicy, true, false, 256, true>(const ScanParams, const GeomConst, const TileGrid, DevState*, const KalmanRecLayers, float* const*, int, const TilePool, const TileAux, const TileWork, unsigned, const ScanParams, const ScanInputs, const Scratch, const TilePool, int32_t*);
EOS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -gline-tables-only -save-temps "$@" -Rpass-analysis=kernel-resource-usage -c one.hip -o one.o 2>&1 | awk '
/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
/ VGPRs:/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v)}
/TotalSGPRs:/ {sg=$0; sub(/.*TotalSGPRs: /,"",sg); sub(/ .*/,"",sg)}
/SGPRs Spill/ {sp=$0; sub(/.*: /,"",sp); sub(/ .*/,"",sp)}
/ScratchSize/ {sc=$0; sub(/.*: /,"",sc); sub(/ .*/,"",sc)}
/Occupancy \[waves/ {oc=$0; sub(/.*: /,"",oc); sub(/ .*/,"",oc)}
/LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l); print "vgpr", v, "sgpr", sg, "sgpr_spill", sp, "scratch", sc, "occ", oc, "lds", l, name}' | while read a v b sg c sp d sc e oc f l name; do echo "vgpr $v sgpr $sg spill $sp scratch $sc occ $oc lds $l $(echo $name | c++filt | cut -c1-70)"; done
