#!/usr/bin/env python3
"""Static check of the kernels' ISA for the two hazards the compiler cannot see through inline assembly:

  * VALU -> MFMA: an inline-assembly vector instruction whose result an MFMA reads as A / B within WAIT_VALU_MFMA wait states
    (the hazard recogniser inserts `s_nop` for its own instructions only).  This is the bug of commit 1e070e2: the sparse block-1 weight
    gradient read stale B operands and passed every isolated kernel test.
  * MFMA -> inline assembly: an assembly instruction (the untracked epilogue stores, lane merges) that reads an accumulator an MFMA
    wrote fewer than WAIT_MFMA_READ wait states earlier (16-pass MFMA: 18; the conv epilogues spend `s_nop 15; s_nop 7` once per tile).

  python tools/asm_hazard_scan.py            # compiles every csrc/*.hip that issues MFMAs to ISA (hipcc -S, no GPU needed)
  python tools/asm_hazard_scan.py file.s ... # scans existing listings

Every instruction counts as one wait state, `s_nop N` as N + 1: conservative for the second check, exact for the first.
Control flow: at every label the scan continues with the history of EVERY branch that targets it next to the fall-through one (loop
back-edges and joins), so the first assembly instruction of a loop body is checked against the last MFMAs of the previous trip.
Exit status 1 and one line per finding when something is found."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

WAIT_VALU_MFMA = 2
WAIT_MFMA_READ = 18
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'exploring_meta_amd', 'csrc')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-S', '--cuda-device-only']
VGPR_FORM = ['-mllvm', '-amdgpu-mfma-vgpr-form=1']
AGPR_FILES = ('policy_sweep.hip', 'wgrad_bf16.hip')          # as in csrc/Makefile


def regs(tok):
    tok = tok.strip().rstrip(',').split()[0] if tok.strip() else ''
    m = re.match(r'[va]\[(\d+):(\d+)\]$', tok)
    if m:
        return {(tok[0], i) for i in range(int(m.group(1)), int(m.group(2)) + 1)}
    m = re.match(r'[va](\d+)$', tok)
    return {(tok[0], int(m.group(1)))} if m else set()


def walk(path, preds=None):
    """One pass over a listing.  `preds`: label -> the instruction histories at the branches that target it (from an earlier pass); at such
    a label the walk continues with EVERY predecessor's history next to the fall-through one -- loop back-edges and joins included -- so
    an assembly instruction at the top of a loop body is checked against the MFMAs at the bottom of the previous trip.  Returns
    (findings, label -> histories at the branches to it)."""
    findings, kernel, inasm = set(), None, False
    hists = [[]]                 # alternative histories: (is_asm, is_mfma, dst registers, text), one entry per wait state
    tails = {}
    KEEP = max(WAIT_MFMA_READ, WAIT_VALU_MFMA) + 6

    def push(entry, n=1):
        for h in hists:
            h.extend([entry] * n)
            if len(h) > 4 * KEEP:
                del h[:len(h) - KEEP]

    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            kernel, hists = m.group(1), [[]]
            continue
        t = line.strip()
        if t.startswith(';;#ASMSTART'):
            inasm = True
            continue
        if t.startswith(';;#ASMEND'):
            inasm = False
            continue
        m = re.match(r'^(\.LBB\w+):', t)
        if m:
            if preds is not None:
                for tail in preds.get((kernel, m.group(1)), []):
                    if not any(h[-KEEP:] == tail[-KEEP:] for h in hists):
                        hists.append(list(tail))
            continue
        if not t or t[0] in '.;' or t.endswith(':'):
            continue
        parts = t.split(None, 1)
        op, ops = parts[0], (parts[1].split(',') if len(parts) > 1 else [])
        if op == 's_nop':
            push((False, False, frozenset(), t), int(ops[0]) + 1 if ops else 1)
            continue
        if op.startswith(('s_cbranch', 's_branch')) and ops:
            tails.setdefault((kernel, ops[0].strip()), []).append(list(max(hists, key=len)[-KEEP:]))
        is_mfma = op.startswith('v_mfma')
        is_store = op.startswith(('buffer_store', 'global_store', 'flat_store', 'ds_write', 'scratch_store'))
        dst = frozenset() if is_store else frozenset(regs(ops[0]) if ops else set())
        src = set()
        for o in (ops if is_store else ops[1:]):
            src |= regs(o)
        for hist in hists:
            if is_mfma:
                ab = set()
                for o in ops[1:3]:
                    ab |= regs(o)
                for back, (h_asm, _, h_dst, h_txt) in enumerate(reversed(hist[-WAIT_VALU_MFMA:])):
                    if h_asm and h_txt.startswith('v_') and (h_dst & ab):
                        findings.add(f'{os.path.basename(path)}: {kernel}: MFMA "{t[:70]}" reads an operand that inline assembly "{h_txt[:50]}" wrote {back} wait states earlier')
                        break
            if inasm and not is_mfma:
                # per source register, the NEAREST earlier writer decides: a register an MFMA wrote and an ordinary instruction has
                # redefined since (accumulator registers are re-used for temporaries once a tile's sums are out) is not a matrix result
                for r in src:
                    for back, (_, h_mfma, h_dst, h_txt) in enumerate(reversed(hist[-WAIT_MFMA_READ:])):
                        if r in h_dst:
                            if h_mfma:
                                findings.add(f'{os.path.basename(path)}: {kernel}: inline assembly "{t[:60]}" reads the result of "{h_txt[:50]}" after {back} wait states')
                            break
        push((inasm, is_mfma, dst, t))
        if len(hists) > 1:       # alternatives that have become equal over the window that matters collapse into one
            uniq = []
            for h in hists:
                if not any(u[-KEEP:] == h[-KEEP:] for u in uniq):
                    uniq.append(h)
            hists = uniq
    return sorted(findings), tails


def scan(path):
    """Two passes: the first collects the history at every branch, the second walks with all predecessors joined in at every label."""
    _, tails = walk(path)
    findings, _ = walk(path, tails)
    return findings


def compile_listing(src, outdir):
    out = os.path.join(outdir, os.path.basename(src)[:-4] + '.s')
    flags = FLAGS + ([] if os.path.basename(src) in AGPR_FILES else VGPR_FORM)
    subprocess.run(['hipcc'] + flags + ['-o', out, src], check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def main(argv):
    if argv:
        listings = argv
    else:
        srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.hip') and '_mfma_' in open(os.path.join(CSRC, f)).read()]
        tmp = tempfile.mkdtemp(prefix='mi_isa_')
        with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
            listings = list(ex.map(lambda s: compile_listing(s, tmp), srcs))
    findings = [f for p in listings for f in scan(p)]
    for f in findings:
        print(f)
    print(f'{len(listings)} listings scanned, {len(findings)} findings')
    return 1 if findings else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
