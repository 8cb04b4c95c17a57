"""Rank-0-first start-up work without a collective.

Long single-rank start-up work -- MIOpen's convolution-kernel search (1.5 to 8 minutes on a fresh node), building the HIP
library -- must not be waited for inside `dist.barrier()`: the waiting ranks sit in an RCCL collective whose watchdog (the
process group's timeout, 10 minutes by default) aborts the whole job when rank 0 takes longer.  A NodeLatch is a marker file in
the node's temp directory instead: rank 0 releases it when it is done, the other ranks of the node poll for it -- no collective
is pending while they wait.  Single node only (like the reference: train.py:36-44 pins MASTER_ADDR to localhost)."""
import os
import tempfile
import time


class NodeLatch:
    def __init__(self, name, directory=None):
        # one job = one launcher: every rank of a node is a child of the same torch.distributed.run agent (or mp.spawn parent)
        key = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
        self.path = os.path.join(directory or tempfile.gettempdir(), f"cnerf_latch_{key}_{name}")

    def release(self):
        tmp = self.path + f".{os.getpid()}"
        with open(tmp, "w") as f:
            f.write(str(time.time()))
        os.replace(tmp, self.path)

    def wait(self, poll_s=0.2, timeout_s=4 * 3600.0):
        t0 = time.monotonic()
        while not os.path.exists(self.path):
            if time.monotonic() - t0 > timeout_s:
                raise TimeoutError(f"rank 0 never released {self.path}")
            time.sleep(poll_s)

    def clear(self):
        try:
            os.remove(self.path)
        except OSError:
            pass


def rank0_first(rank, name, work, directory=None):
    """Runs work() on rank 0, then on no other rank; ranks != 0 return once rank 0 has finished (or raise if it failed:
    the marker then never appears and the launcher tears the job down).  Returns work()'s result on rank 0, None elsewhere."""
    latch = NodeLatch(name, directory)
    if rank == 0:
        latch.clear()
        out = work()
        latch.release()
        return out
    latch.wait()
    return None
