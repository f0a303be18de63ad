"""Run an unmodified reference script against this package:

    python -m locotouch_amd.compat.run_reference /path/to/LocoTouch/locotouch/scripts/train.py --task Isaac-...-v1 --num_envs 4096 --headless

Adds the checkout's package roots to sys.path (the script's own directory for `import cli_args`, the repo root for
`import locotouch`), installs locotouch_amd.compat.runtime and executes the script as `__main__`.
"""
from __future__ import annotations

import os
import runpy
import sys


def main() -> None:
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    script = os.path.abspath(sys.argv[1])
    scripts_dir = os.path.dirname(script)
    repo_root = os.path.dirname(os.path.dirname(scripts_dir))  # .../locotouch/scripts/x.py -> checkout root
    sys.dont_write_bytecode = True
    for p in (scripts_dir, repo_root):
        if p not in sys.path:
            sys.path.insert(0, p)
    from locotouch_amd.compat import runtime

    runtime.install()
    sys.argv = [script] + sys.argv[2:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
