#!/usr/bin/env python3
"""Basic blocks of one kernel in `hipcc -S --offload-device-only -gline-tables-only` output: per block the VALU / v_mov /
SALU / memory instruction counts, the waits, the branch targets and the first source lines, in layout order. Used to
follow the path a search-mode step takes through the compiled kernel (which waits and copies sit on it).

usage: asm_blocks.py kernel-symbol-substring [file.s]      (default file: /tmp/demod_g.s)"""
import re,sys
kern=sys.argv[1]
inside=False
blocks=[]
cur=None
loc=None
for l in open(sys.argv[2] if len(sys.argv) > 2 else '/tmp/demod_g.s'):
    t=l.strip()
    m=re.match(r'^(_Z\w+):',t)
    if m:
        inside = kern in m.group(1); continue
    if not inside: continue
    m=re.match(r'^(\.LBB\d+_\d+):',t)
    if m or t.startswith('; %bb.'):
        name = m.group(1) if m else t.split(':')[0][2:]
        cur={'name':name,'valu':0,'vmov':0,'salu':0,'vmem':0,'lds':0,'locs':[],'br':[], 'wait':0}
        blocks.append(cur); continue
    if cur is None: continue
    m=re.match(r'\.loc\s+(\d+)\s+(\d+)\s+\d+.*?; (\S+)',t)
    if m:
        if len(cur['locs'])<2: cur['locs'].append(m.group(3))
        continue
    if re.match(r'^v_',t):
        cur['valu']+=1
        if t.startswith('v_mov') : cur['vmov']+=1
    elif re.match(r'^s_',t):
        cur['salu']+=1
        if t.startswith('s_cbranch') or t.startswith('s_branch'): cur['br'].append(t.split()[0][2:]+'>'+t.split()[1])
        if t.startswith('s_waitcnt'): cur['wait']+=1
    elif re.match(r'^(global_|buffer_|flat_|scratch_)',t): cur['vmem']+=1
    elif t.startswith('ds_'): cur['lds']+=1
for b in blocks:
    print("%-12s v=%3d mov=%3d s=%3d m=%2d w=%d %s %s"%(b['name'],b['valu'],b['vmov'],b['salu'],b['vmem'],b['wait'],' '.join(b['br']),' '.join(x.replace('csrc/','') for x in b['locs'])))
