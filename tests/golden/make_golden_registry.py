"""Every environment id the reference registers with the keyword arguments ``make(id)`` ends up passing to the env class
(``fluidgym/__init__.py:28-352``: ``register(id, entry_point, defaults=<FAMILY>_DEFAULT_CONFIG, **overrides)``; the default
config dictionaries live next to the env classes).  Extracted HERE with ``ast`` -- values are stored as the text of their
literal (``torch.float32`` stays that text); no source is copied.

    python tests/golden/make_golden_registry.py        ->  tests/golden/reference_registry.json
"""
import ast
import glob
import json
import os

REF = "/root/reference/src/fluidgym"
OUT = os.path.dirname(os.path.abspath(__file__))


def default_configs():
    """module-level ``NAME_DEFAULT_CONFIG = {...}`` dictionaries of the env modules (a dict may extend another: ``{**BASE, ...}``)"""
    raw = {}
    for path in glob.glob(f"{REF}/envs/**/*.py", recursive=True):
        tree = ast.parse(open(path).read())
        for n in tree.body:
            tgt, val = None, None
            if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name):
                tgt, val = n.targets[0].id, n.value
            elif isinstance(n, ast.AnnAssign) and isinstance(n.target, ast.Name) and n.value is not None:
                tgt, val = n.target.id, n.value
            if tgt and tgt.endswith("_DEFAULT_CONFIG") and isinstance(val, ast.Dict):
                raw[tgt] = val
    done = {}

    def resolve(name):
        if name in done:
            return done[name]
        out = {}
        for k, v in zip(raw[name].keys, raw[name].values):
            if k is None:                       # ** expansion of another config
                out.update(resolve(ast.unparse(v)))
            else:
                out[ast.literal_eval(k)] = ast.unparse(v)
        done[name] = out
        return out

    return {name: resolve(name) for name in raw}


def main():
    cfgs = default_configs()
    tree = ast.parse(open(f"{REF}/__init__.py").read())
    envs = {}
    for n in ast.walk(tree):
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id == "register":
            kw = {k.arg: k.value for k in n.keywords}
            env_id = ast.literal_eval(kw.pop("id"))
            entry = ast.unparse(kw.pop("entry_point"))
            defaults = ast.unparse(kw.pop("defaults"))
            merged = dict(cfgs[defaults])
            merged.update({k: ast.unparse(v) for k, v in kw.items()})
            envs[env_id] = {"entry_point": entry, "defaults": defaults, "kwargs": merged}
    with open(os.path.join(OUT, "reference_registry.json"), "w") as f:
        json.dump(envs, f, indent=1, sort_keys=True)
    print(len(envs), "ids;", sorted({v["entry_point"] for v in envs.values()}))


if __name__ == "__main__":
    main()
