"""Static scan of a built libvpmi for the gfx950 hazard of DESIGN.md section 8 ("packed-f32 read of a freshly loaded pair"):

    global_load / buffer_load  v[a:b]  ...           (more loads in flight behind it)
    s_waitcnt vmcnt(N), N > 0                         (counted: the OLDER loads have returned, younger ones are still in flight)
    v_pk_{fma,mul,add}_f32 ..., v[a:a+1], ...         (a packed-f32 VALU reads the loaded pair)

Measured on MI355X (tools/stress_canary.py, profiles/r04_corun_canary.log): under an MFMA-heavy neighbour on the same SIMD the low
register of the pair is read STALE in lanes 48-63 (21-25 % of launches); the same loop with four v_fma_f32, or with vmcnt(0)
before the packed ops, never fails.  The scan walks every kernel of the library linearly, keeps the vmcnt queue the way the
hardware does (loads and stores retire in order), and reports packed-f32 instructions with a source register that came back under a
counted wait and has not been rewritten since.  Branch targets keep the state (over-approximation: loops are walked once).

Round 4 outcome: the narrow pattern above is what se_gate showed, but the Fbank kernel failed under the same neighbours with its packed
ops elsewhere, and only a build with NO packed-f32 instruction at all (-target-feature -packed-fp32-ops, build.py) passed every
screen.  The gate of this scan is therefore the COUNT: a product library must contain zero v_pk_{fma,mul,add}_f32.

Usage: python tools/isa_scan.py [path/to/libvpmi.so] [--list]      exit code 1 when the library contains packed-f32 VALU instructions
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
PK = ('v_pk_fma_f32', 'v_pk_mul_f32', 'v_pk_add_f32')


def regs(tok):
    """'v[14:17]' -> {14..17}; 'v3' -> {3}; anything else -> empty"""
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix='isa_scan_')
    work = os.path.join(tmp, os.path.basename(lib))
    os.symlink(os.path.abspath(lib), work)
    subprocess.run([f'{LLVM}/llvm-objdump', '--offloading', work], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = []
    for f in sorted(os.listdir(tmp)):
        if 'amdgcn' in f:
            out.append(subprocess.run([f'{LLVM}/llvm-objdump', '-d', '--no-show-raw-insn', os.path.join(tmp, f)],
                                      stdout=subprocess.PIPE, text=True).stdout)
    return '\n'.join(out)


def scan(text):
    """Two walks per kernel, the second starting from the state the first one ended with: a load issued at the bottom of a loop
    and consumed at its top (software prefetch) is then seen by the consumer."""
    kernels = collections.OrderedDict()
    kernel = None
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
        if m:
            name = m.group(1)
            if not name.startswith('L') and not name.startswith('.L'):        # a new function, not a local label
                kernel = name
                kernels.setdefault(kernel, [])
            continue
        line = line.split('//')[0].strip()
        if line and kernel is not None:
            kernels[kernel].append(line)
    hits = collections.OrderedDict()
    npk = collections.Counter()
    for kernel, lines in kernels.items():
        queue, counted = [], set()
        for walk in range(2):
            for line in lines:
                queue, counted = step(kernel, line, queue, counted, hits, npk, walk)
    for k in hits:
        hits[k] = list(dict.fromkeys(hits[k]))
    return hits, npk


def step(kernel, line, queue, counted, hits, npk, walk):
    if True:
        parts = line.split(None, 1)
        op = parts[0]
        ops = [t.strip() for t in re.split(r',\s*(?![^\[]*\])', parts[1])] if len(parts) > 1 else []
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', line)
            if m:
                n = int(m.group(1))
                done = queue[:len(queue) - n] if n < len(queue) else []
                queue = queue[len(queue) - n:] if n else []
                if n == 0:
                    counted = set()
                else:
                    for d in done:
                        counted |= d
            return queue, counted
        is_load = re.match(r'(global|buffer|flat|scratch)_load', op) and ' lds' not in line
        is_store = re.match(r'(global|buffer|flat|scratch)_(store|atomic)', op)
        if is_load:
            dst = regs(ops[0].split()[0]) if ops else set()
            counted -= dst
            queue.append(dst)
            return queue, counted
        if is_store or (re.match(r'(global|buffer)_load', op) and ' lds' in line):
            queue.append(set())
            return queue, counted
        if op in PK:
            npk[kernel] += 1 if walk == 0 else 0
            src = set()
            for t in ops[1:]:
                src |= regs(t.split()[0])
            bad = src & counted
            if bad:
                hits.setdefault(kernel, []).append(line)
        # any other instruction that writes VGPRs takes them out of the set
        if ops and op.startswith(('v_', 'ds_read', 'ds_bpermute', 'ds_swizzle')):
            counted -= regs(ops[0].split()[0])
    return queue, counted


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    lib = args[0] if args else os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd', 'lib', 'libvpmi.so')
    text = disassemble(lib)
    hits, npk = scan(text)
    demangle = lambda n: subprocess.run(['c++filt', n], stdout=subprocess.PIPE, text=True).stdout.strip()
    print(f'{lib}: {sum(npk.values())} packed-f32 VALU instructions in {len(npk)} kernels; '
          f'{sum(len(v) for v in hits.values())} read a pair that came back under a counted vmcnt, in {len(hits)} kernels')
    for k, v in hits.items():
        print(f'  {demangle(k)[:150]}: {len(v)}   e.g. {v[0]}')
    if '--list' in sys.argv:
        print('kernels with packed-f32 VALU instructions:')
        for k, n in npk.most_common():
            print(f'  {n:5d}  {demangle(k)[:150]}')
    return 1 if sum(npk.values()) else 0


if __name__ == '__main__':
    sys.exit(main())
