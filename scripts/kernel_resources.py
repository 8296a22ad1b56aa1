#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: VGPRs / AGPRs / scratch / occupancy per kernel.
usage: kernel_resources.py FILE [substring ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filt = sys.argv[2:]
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
names = [b.split('\n')[0].strip() for b in blocks]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
for b, nm in zip(blocks, dem):
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    if filt and not all(f in nm for f in filt):
        continue
    print('vgpr %3d agpr %3d scratch %4d occ %d lds %6d  %s' % (g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'),
          g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]'), nm[:230]))
