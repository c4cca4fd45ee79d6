#!/usr/bin/env python3
"""Issue-slot cost model of the trace kernel's ISA: per basic block, the number of VALU instructions
and their cost in units of one full-rate VALU instruction (rates measured by tools/ubench/valu_rates.hip
on MI355X: f32 add/sub/mul/fma, u32 add/sub, and/or/xor, right shifts and mov with VGPR, literal or
inline operands = 1; the same with an SGPR operand, left shifts, compares, selects, conversions,
min/max, integer multiplies, three-operand integer ops, DPP and the division helpers = 1.8;
packed f32 = 2; rcp/sqrt/rsq = 3.5).  Usage: python tools/isa_cost.py /tmp/rt_kernels.s [kernel-name-substring]"""
import re
import sys

FULL = ('v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_mul_f32', 'v_fma_f32', 'v_fmac_f32', 'v_mac_f32', 'v_and_b32',
        'v_or_b32', 'v_xor_b32', 'v_lshrrev_b32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32',
        'v_mov_b32', 'v_not_b32', 'v_ashrrev_i32')
TRANS = ('v_rcp', 'v_sqrt', 'v_rsq', 'v_exp', 'v_log')


SGPR_OPERAND = re.compile(r'[ ,]s\d+\b|[ ,]s\[\d+:\d+\]|\bvcc\b')


def cost(op, operands=''):
    if op.startswith(TRANS):
        return 3.5
    if op.startswith('v_pk_'):
        return 2.0
    if op.startswith(FULL) and not SGPR_OPERAND.search(' ' + operands):
        return 1.0
    return 1.8


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else 'rt_trace_kernelILi5ENS0_7NoStats'
    text = open(path).read().split('\n')
    inside = False
    blocks = []
    cur = None
    for i, line in enumerate(text, 1):
        if not inside:
            if re.match(r'^_Z\w*' + re.escape(want) + r'\w*:', line):
                inside = True
                cur = {'name': 'entry', 'line': i, 'n': 0, 'units': 0.0, 'ops': {}}
                blocks.append(cur)
            continue
        if 's_endpgm' in line:
            break
        m = re.match(r'^(\.LBB\d+_\d+|; %bb\.\d+):?', line)
        if m:
            cur = {'name': m.group(1), 'line': i, 'n': 0, 'units': 0.0, 'ops': {}}
            blocks.append(cur)
            continue
        tok = line.strip().split()
        if not tok:
            continue
        op = tok[0]
        if op.startswith('v_'):
            cur['n'] += 1
            cur['units'] += cost(op, ' '.join(tok[1:]))
            short = op.replace('_e32', '').replace('_e64', '')
            cur['ops'][short] = cur['ops'].get(short, 0) + 1
    total = 0.0
    for b in blocks:
        if b['n'] == 0:
            continue
        total += b['units']
        top = sorted(b['ops'].items(), key=lambda kv: -kv[1])[:6]
        print("%-10s L%-5d n=%-3d units=%-6.1f %s" % (b['name'], b['line'], b['n'], b['units'],
                                                      ' '.join('%s:%d' % kv for kv in top)))
    print("total static units: %.1f" % total)


if __name__ == '__main__':
    main()
