#!/usr/bin/env python3
"""Development tool: walks one kernel's disassembly along a chosen path and counts the instructions on it by class.
usage: tools/isa_trace.py <kernel.s> <start line> <stop line> [taken-branch lines, comma separated]
The disassembly is `llvm-objdump -d` of the code object (tools/isa_dump.sh); conditional branches fall through unless their
line number is listed; unconditional branches are followed.  Prints per-class counts and the lines visited as ranges."""
import re, sys, collections

def cls(op):
    if op.startswith('v_'):
        if 'f64' in op: return 'v_f64'
        if op.startswith('v_cvt'): return 'v_cvt'
        if op.startswith(('v_mov', 'v_accvgpr')): return 'v_mov'
        if op.startswith(('v_cmp', 'v_cndmask')): return 'v_cmp/cnd'
        if op.startswith(('v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_fmaak', 'v_fmamk')): return 'v_f32 mul/add'
        if op.startswith(('v_fmac_f32', 'v_fma_f32')): return 'v_f32 fma'
        if op.startswith(('v_min', 'v_max', 'v_med3', 'v_fract')): return 'v_min/max/fract'
        if op.startswith(('v_readfirstlane', 'v_readlane', 'v_writelane')): return 'v_lane'
        if op.startswith(('v_rcp', 'v_div')): return 'v_div/rcp'
        return 'v_int/other'
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith(('s_cbranch', 's_branch')): return 's_branch'
    if op.startswith(('s_nop', 's_setprio')): return 's_nop/setprio'
    if op.startswith('s_'): return 's_alu'
    if op.startswith('ds_'): return 'ds'
    if op.startswith(('global_', 'buffer_', 'flat_')): return 'vmem'
    return 'other'

def main():
    lines = open(sys.argv[1]).read().split('\n')
    start, stop = int(sys.argv[2]), int(sys.argv[3])
    taken = set(int(x) for x in sys.argv[4].split(',')) if len(sys.argv) > 4 and sys.argv[4] else set()
    addr = {}
    for i, l in enumerate(lines):
        m = re.search(r'// ([0-9A-F]{12}):', l)
        if m: addr[int(m.group(1), 16)] = i + 1
    c = collections.Counter(); visited = []; n = start; steps = 0
    while (n != stop or steps == 0) and steps < 20000:
        steps += 1
        l = lines[n - 1]
        m = re.match(r'\s+(\S+)', l)
        if not m: n += 1; continue
        op = m.group(1); c[cls(op)] += 1; visited.append(n)
        if op.startswith(('s_cbranch', 's_branch')):
            off = int(re.search(r'branch\S*\s+(\d+)', l).group(1))
            if off >= 32768: off -= 65536
            a = int(re.search(r'// ([0-9A-F]{12}):', l).group(1), 16)
            tgt = addr[a + 4 + 4 * off]
            if op == 's_branch' or n in taken: n = tgt; continue
        n += 1
    tot = sum(c.values()); valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print('instructions on the path: %d, VALU %d' % (tot, valu))
    for k, v in sorted(c.items(), key=lambda x: -x[1]): print('  %-16s %5d' % (k, v))
    r = []; s = visited[0]; p = s
    for x in visited[1:]:
        if x != p + 1: r.append((s, p)); s = x
        p = x
    r.append((s, p))
    print('lines:', ' '.join('%d-%d' % x for x in r))

if __name__ == '__main__':
    main()
