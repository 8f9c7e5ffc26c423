"""torchrun worker of test_cli_two_ranks_gloo: one rank of the 2-rank CPU (gloo) pipeline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from tests.test_cli_e2e import run_rank
    run_rank(sys.argv[1])
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()
