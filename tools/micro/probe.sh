#!/bin/sh
# usage: tools/micro/probe.sh [-DPROBE_EPI=n]   -> register table of the instances in regprobe.hip
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Rpass-analysis=kernel-resource-usage -I../../tmlqcd_amd/csrc "$@" -c regprobe.hip -o /dev/null 2> /tmp/ru.txt
python3 ../check_resources.py --table /tmp/ru_table.txt /tmp/ru.txt > /dev/null 2>&1
grep "hop_kernel\|exterior\|pack" /tmp/ru_table.txt | cut -c1-60,96-150
grep -i "error" /tmp/ru.txt | head
