#!/usr/bin/env python3
"""bench.py with the device layer replaced by tests/fake_device.py: a DRY RUN of the script's logic on the CPU (no timing means
anything).  Same command line as bench.py; `--gpus N` starts N gloo ranks of THIS wrapper.
    python tests/bench_dry_run.py --gpus 2 --workload awb_mixed --scaling strong --awb-clips 40 --awb-durations 12 --no-cpu"""
import os
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    import bench
    import fake_device
    if os.environ.get("CRI_TEST_HOSTWAVE") == "1":             # the real batch.Job on the emulated kernels (tests/hostwave), not the oracle-backed double
        fake_device.install_emulated(bench)
    else:
        fake_device.install(bench)

    def relaunch(args):                                        # bench.relaunch_under_torchrun, on this wrapper
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    bench.relaunch_under_torchrun = relaunch
    bench.main()


if __name__ == "__main__":
    main()
