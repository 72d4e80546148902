#!/bin/bash
# usage: tools/kres_short.sh file.hip  -- one line per kernel: template arguments, VGPRs, scratch bytes per lane
KFLAGS="$KFLAGS" $(dirname $0)/kres.sh $1 2>&1 | sed 's/Function Name: _ZN[0-9]*_GLOBAL__N_1[0-9]*//; s/EEEv[0-9]*pmi_[a-z_]*args//; s/Occupancy.*//' | awk '{print $1, "v" $3, "s" $6}' | sort | paste - - - | sed 's/\t/   |   /g'
