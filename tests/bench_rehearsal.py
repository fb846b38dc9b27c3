"""CPU rehearsal of bench.py's N > 1 control flow: the SAME bench.main() -- argument parsing,
rank / world bookkeeping, process-group setup, barrier + max-over-ranks timing, the owners assertion, the strong-
scaling leg, JSON emission on rank 0 only, non-zero exit of a failing rank -- on a platform object whose device is the
CPU, whose group is gloo and whose index pieces are the oracle-backed doubles of tests/test_sharded_gloo.py.
Launched by tests/test_bench_rehearsal.py as
    python -m torch.distributed.run --nproc-per-node 2 ... tests/bench_rehearsal.py --gpus 2 ...
This file is test infrastructure: the product and bench.py never import it."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


class CpuRehearsalPlatform:
    name = "cpu-rehearsal"

    def __init__(self, local_rank: int):
        import torch

        self.local_rank = local_rank
        self.dev = torch.device("cpu")
        self.reduce_dev = self.dev

    def init_group(self):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")

    def barrier(self):
        import torch.distributed as dist

        dist.barrier()

    def sync(self):
        pass

    def stream(self) -> int:
        return 0

    def make_sharded(self, dim: int):
        from test_sharded_gloo import _FakeLocal, _merge

        from claude_semantic_search_amd.sharded import ShardedFlatIndex

        fail_rank = os.environ.get("CSS_REHEARSAL_FAIL_RANK")
        if fail_rank is not None and int(os.environ.get("RANK", "0")) == int(fail_rank):
            raise RuntimeError("injected failure (CSS_REHEARSAL_FAIL_RANK)")
        return ShardedFlatIndex(dim, 0, index_factory=lambda: _FakeLocal(dim, 0), merge=_merge(0))


if __name__ == "__main__":
    import bench

    bench.main(sys.argv[1:], platform_factory=CpuRehearsalPlatform)
