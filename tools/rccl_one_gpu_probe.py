"""Can RCCL (torch.distributed backend "nccl") run TWO ranks on ONE GPU?  (the builder's box has one MI355X, the
collectives of the data-parallel step have only ever run over gloo.)  Two processes, both on cuda:0, one all-reduce of 26.9 MB in the
step's own chunk list (ppvector/train/step.py: reduce_chunks) -- prints the outcome, or RCCL's refusal, per rank.

    python tools/rccl_one_gpu_probe.py            # exit code 0 either way; the answer is in the output
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    import torch
    import torch.distributed as dist
    try:
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
        from ppvector.train.step import reduce_chunks
        n = 6_730_000
        g = torch.full((n,), float(rank + 1), device='cuda')
        chunks = reduce_chunks(n)
        torch.cuda.synchronize()
        t0 = time.time()
        works = [dist.all_reduce(g[lo:hi], async_op=True) for lo, hi in chunks]
        for w in works:
            w.wait()
        torch.cuda.synchronize()
        dt = time.time() - t0
        ok = bool((g == float(sum(range(1, world + 1)))).all())
        q.put((rank, 'ok' if ok else 'WRONG SUM', f'{len(chunks)} chunks, {n * 4 / 1e6:.1f} MB in {dt * 1e3:.2f} ms (first call, includes set-up)'))
        dist.destroy_process_group()
    except Exception as e:                                    # noqa: BLE001 -- the refusal IS the result
        q.put((rank, 'refused', f'{type(e).__name__}: {str(e)[:600]}'))


def main():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 41000 + os.getpid() % 2000
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = []
    t0 = time.time()
    while len(got) < 2 and time.time() - t0 < 120:
        try:
            got.append(q.get(timeout=5))
        except Exception:                                     # noqa: BLE001
            if not any(p.is_alive() for p in procs):
                break
    for p in procs:
        p.join(10)
        if p.is_alive():
            p.kill()
    print('# RCCL with two ranks on one MI355X (both on cuda:0)')
    for r in sorted(got):
        print(f'rank {r[0]}: {r[1]} -- {r[2]}')
    if len(got) < 2:
        print(f'{2 - len(got)} rank(s) gave no answer within 120 s (hung or died): exit codes {[p.exitcode for p in procs]}')


if __name__ == '__main__':
    main()
