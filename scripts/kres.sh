#!/bin/bash
# compact per-kernel resource table from `make asm` remarks: name vgpr sgpr scratch occupancy lds
# usage: scripts/kres.sh [grep-pattern]
make -C fastdem_amd/csrc asm > build/kres.log 2>&1; awk '
/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
/ VGPRs:/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v)}
/TotalSGPRs:/ {sg=$0; sub(/.*TotalSGPRs: /,"",sg); sub(/ .*/,"",sg)}
/ScratchSize/ {sc=$0; sub(/.*: /,"",sc); sub(/ .*/,"",sc)}
/Occupancy \[waves/ {oc=$0; sub(/.*: /,"",oc); sub(/ .*/,"",oc)}
/LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ .*/,"",l); print v, sg, sc, oc, l, name}
' build/kres.log | while read v sg sc oc l name; do echo "$v $sg $sc $oc $l $(echo $name | c++filt | cut -c1-110)"; done | grep -E "${1:-.}"
