"""`run-smart-tree` entry point (reference smart_tree/cli.py:10-26) without hydra: the YAML `_target_`
tree is instantiated by a small recursive loader, `+path=...` / `+directory=...` and `a.b=value`
overrides are accepted on the command line."""
from __future__ import annotations

import importlib
import sys
from pathlib import Path

import yaml


def instantiate(node):
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    if not isinstance(node, dict):
        return node
    kwargs = {k: instantiate(v) for k, v in node.items() if k != "_target_"}
    if "_target_" not in node:
        return kwargs
    module, _, name = node["_target_"].rpartition(".")
    return getattr(importlib.import_module(module), name)(**kwargs)


def load_config(overrides=()):
    cfg = yaml.safe_load((Path(__file__).resolve().parent / "conf" / "pipeline.yaml").read_text())
    for item in overrides:
        key, _, value = item.lstrip("+").partition("=")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(value)
    return cfg


def main(argv=None):
    cfg = load_config(argv if argv is not None else sys.argv[1:])
    pipeline = instantiate(cfg["pipeline"])
    if "path" in cfg:
        pipeline.process_cloud(Path(cfg["path"]))
    elif "directory" in cfg:  # every entry of the directory, as cli.py:22-23 (sorted; files load_cloud cannot read are named and skipped)
        for p in sorted(Path(cfg["directory"]).iterdir()):
            if p.is_file() and p.suffix in (".npz", ".ply"):
                pipeline.process_cloud(p)
            elif p.is_file():
                print(f"skipping {p}: not a point cloud format this build reads (.npz / .ply)")
    else:
        print("Please supply a path or directory.")


if __name__ == "__main__":
    main()
