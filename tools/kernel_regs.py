#!/usr/bin/env python
"""Register / spill / scratch table of every kernel in a --save-temps .s file (gfx950 code object metadata)."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
meta = s[s.index('amdhsa.kernels:'):]
for blk in re.split(r'\n  - \.agpr_count:', meta)[1:]:
    nm = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if pat not in nm:
        continue
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    print('%-75s agpr %3s vgpr %3s spill %3s scratch %4s lds %6s' % (nm[:75], blk.split('\n')[0].strip(), g('vgpr_count'), g('vgpr_spill_count'),
          g('private_segment_fixed_size'), g('group_segment_fixed_size')))
