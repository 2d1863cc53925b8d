"""One rank of the 2-process shared-card rehearsal (tests/test_gpu_dist.py): `lemon_amd.run_lemon.main` on a loop fixture's
planted model / data under WORLD_SIZE=2, LEMON_DIST_BACKEND=gloo (RCCL needs distinct devices; both ranks use cuda:0)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Patch:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def main():
    case_name, out_dir, data_dir = sys.argv[1:4]
    from pathlib import Path
    from tests import planted
    from tests.loopfx import LoopCase
    from lemon_amd.run_lemon import main as run
    c = LoopCase(case_name)
    extra = planted.install(c, _Patch(), Path(data_dir) / f"rank{os.environ.get('RANK', '0')}")
    return run(["--output_dir", out_dir] + c.argv + extra)


if __name__ == "__main__":
    sys.exit(main())
