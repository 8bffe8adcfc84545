"""Generate tests/golden/*.npz by running the REFERENCE modules (build container only).

    python tests/golden/make_golden.py [/root/reference]

The reference's ``vqvae/vqvae.py`` is imported from where it lies (never copied); each case of
``tests/golden_cases.py`` is driven through the shared runner and its outputs are stored.  The GPU box
never sees /root/reference -- it only reads the .npz files written here.
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_cases as gc  # noqa: E402


def main():
    ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    sys.path.insert(0, ref_root)
    from vqvae import vqvae as ref  # the reference implementation itself

    torch.manual_seed(0)            # only touches the modules' own init draws, which are overwritten
    torch.set_num_threads(8)

    def factory(cls, **kw):
        return getattr(ref, cls)(**kw)

    meta = {"torch": torch.__version__, "threads": torch.get_num_threads(), "cases": {}}
    only = set(sys.argv[2:])
    for name, case in gc.CASES.items():
        if only and name not in only:
            continue
        res = gc.run_case(case, factory, "cpu")
        gc.save_case(name, res)
        meta["cases"][name] = {k: list(v.shape) for k, v in res.items()}
        print(name, "ok", {k: float(v) for k, v in res.items() if k.endswith("_loss")})
    if not only:
        with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
