"""TEST INFRASTRUCTURE: `bench.py` with the product's C-ABI calls routed to the CPU emulator (oracle/abi_emul.py), for the host-logic
tests of the launcher and of the DDP path on a box without a GPU:

    python tests/bench_emulated.py --gpus 2 --device cpu --config mini --steps 2 --warmup 1

bench.py itself never imports the emulator (the product path has no CPU implementation); this wrapper is what `bench.launch_ranks`
re-executes per rank when it was the script that was started, so every rank installs the emulator before `bench.main()`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "4")))

import bench  # noqa: E402
import mtt_amd  # noqa: E402
from oracle import abi_emul  # noqa: E402

mtt_amd.ops.call = abi_emul.call
mtt_amd.ops.clear_pack_cache()

if __name__ == "__main__":
    bench.main()
