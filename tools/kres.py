"""Diagnostic: registers / scratch / spills per kernel from `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr saved to a file)."""
import re, subprocess, sys
t = open(sys.argv[1]).read()
for b in t.split('Function Name: ')[1:]:
    name = b.split()[0]
    g = lambda k: (re.search(k + r': (\d+)', b) or [None, '?'])[1]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'\(BatchParams.*', '', dn)
    print(f"{dn[:70]:70s} VGPR {g('VGPRs'):>3} SGPR {g('SGPRs'):>3} scratch {g('ScratchSize .bytes/lane.'):>3} occ {g('Occupancy .waves/SIMD.')} spillS {g('SGPRs Spill'):>3} spillV {g('VGPRs Spill'):>2} LDS {g('LDS Size .bytes/block.')}")
