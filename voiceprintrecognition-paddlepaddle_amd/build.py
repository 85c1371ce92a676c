"""Build libvpmi.so (gfx950) in-tree with hipcc.  Usage: python build.py [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libvpmi.so')
SOURCES = ['api.hip', 'fbank.hip', 'melspec.hip', 'conv_gemm.hip', 'conv_gemm_bf16.hip', 'conv_gemm256.hip', 'conv_gemm_f32.hip', 'conv_gemm_x3.hip', 'small_ops.hip', 'head.hip', 'head_tiled.hip', 'losses.hip', 'res2_chain.hip', 'res2_x3.hip', 'asp_x3.hip', 'asp_fused.hip', 'asp_utt.hip', 'ecapa.hip', 'campplus.hip', 'cam_block.hip', 'fcm_conv.hip', 'pointwise.hip', 'resnet_se.hip', 'eres2net.hip', 'augment.hip', 'train_ops.hip', 'wgrad_tr.hip', 'res2_train.hip', 'se_train.hip', 'cam_train.hip']
# -packed-fp32-ops: NO v_pk_{fma,mul,add}_f32 anywhere in the library.  On gfx950 a packed-f32 VALU instruction that reads registers a
# vector-memory load has just returned can see stale data in lanes 48-63 when an MFMA-heavy wave of another kernel shares its SIMD
# (reproducer tools/canary.hip + tools/stress_canary.py; tests/test_isa_cpu.py keeps the count at zero).  The
# flag is a device target feature; the host pass prints "not a recognized feature" for it, filtered below.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'vpmi.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, s.replace('.hip', '.o'))
        objs.append(o)
        cmd = [hipcc()] + FLAGS + ['-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {s}:\n{out}')
        out = '\n'.join(l for l in out.splitlines() if 'is not a recognized feature for this target' not in l)
        if verbose and out.strip():
            print(out)
    cmd = [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
