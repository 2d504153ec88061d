#!/bin/bash
# kernel resource usage of one translation unit of libferhip: tools/kres.sh fer_me.hip
cd "$(dirname "$0")/../h264-fer_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $EXTRA -c "$1" -o /tmp/kres_tmp.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*remark: ([A-Za-z ]+): ([^ ]+).*/\1=\2/' | paste - - - - - -
