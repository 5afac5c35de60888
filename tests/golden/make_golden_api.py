"""Public API surface of the reference's classes on the drop-in boundary, extracted HERE from the reference's
sources with ``ast`` (names, parameters, defaults -- no source text is stored): ``FluidEnv`` (``envs/fluid_env.py``),
``ParallelFluidEnv`` (``envs/parallel_env.py``), the env-side ``Simulation`` (``simulation/simulation.py``), ``EnvMode`` /
``FluidEnvLike`` (``types.py``) and the registry functions (``registry.py``).

    python tests/golden/make_golden_api.py        ->  tests/golden/reference_api.json

``tests/test_api_surface.py`` holds ``fluidgym_amd``'s classes against it.
"""
import ast
import json
import os

REF = "/root/reference/src/fluidgym"
OUT = os.path.dirname(os.path.abspath(__file__))


def _params(fn: ast.FunctionDef):
    a = fn.args
    pos = list(a.posonlyargs) + list(a.args)
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    out = []
    for arg, d in zip(pos, defaults):
        if arg.arg in ("self", "cls"):
            continue
        out.append({"name": arg.arg, "kind": "positional", "has_default": d is not None, "default": None if d is None else ast.unparse(d)})
    if a.vararg:
        out.append({"name": a.vararg.arg, "kind": "var_positional", "has_default": False, "default": None})
    for arg, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append({"name": arg.arg, "kind": "keyword_only", "has_default": d is not None, "default": None if d is None else ast.unparse(d)})
    if a.kwarg:
        out.append({"name": a.kwarg.arg, "kind": "var_keyword", "has_default": False, "default": None})
    return out


def _decorators(fn):
    return [ast.unparse(d) for d in fn.decorator_list]


def describe_class(path, name):
    tree = ast.parse(open(path).read())
    cls = next(n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == name)
    methods, properties, setters, attrs = {}, [], [], []
    for n in cls.body:
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef)):
            decs = _decorators(n)
            if "property" in decs:
                properties.append(n.name)
            elif any(d.endswith(".setter") for d in decs):
                setters.append(n.name)
            elif not n.name.startswith("_") or n.name in ("__init__", "__len__"):
                methods[n.name] = {"params": _params(n), "abstract": "abstractmethod" in decs,
                                   "static": "staticmethod" in decs, "classmethod": "classmethod" in decs}
        elif isinstance(n, ast.AnnAssign) and isinstance(n.target, ast.Name) and not n.target.id.startswith("_"):
            attrs.append(n.target.id)
        elif isinstance(n, ast.Assign):
            for t in n.targets:
                if isinstance(t, ast.Name) and not t.id.startswith("_"):
                    attrs.append(t.id)
    return {"file": os.path.relpath(path, REF), "bases": [ast.unparse(b) for b in cls.bases], "methods": methods,
            "properties": sorted(p for p in properties if not p.startswith("_")), "setters": sorted(set(setters)), "class_attributes": attrs}


def describe_functions(path):
    tree = ast.parse(open(path).read())
    return {n.name: {"params": _params(n)} for n in tree.body if isinstance(n, ast.FunctionDef) and not n.name.startswith("_")}


def env_class_surfaces(entry_points):
    """effective public surface of each registered env class: members of the class and of its bases inside the reference's env
    package (bases resolved by name), constructor parameters of the class itself"""
    import glob

    classes = {}
    for path in glob.glob(f"{REF}/envs/**/*.py", recursive=True):
        for n in ast.parse(open(path).read()).body:
            if isinstance(n, ast.ClassDef):
                classes[n.name] = (path, n)
    out = {}
    for name in entry_points:
        methods, props, order = {}, set(), []
        todo = [name]
        while todo:
            c = todo.pop(0)
            if c not in classes or c in order:
                continue
            order.append(c)
            todo += [ast.unparse(b) for b in classes[c][1].bases]
        for c in reversed(order):                    # most derived last: overrides win
            d = describe_class(classes[c][0], c)
            methods.update(d["methods"])
            props.update(d["properties"])
        consts = {}
        for c in reversed(order):                    # class-level constants with literal values (most derived wins)
            for n in classes[c][1].body:
                tgt, val = None, None
                if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name):
                    tgt, val = n.targets[0].id, n.value
                elif isinstance(n, ast.AnnAssign) and isinstance(n.target, ast.Name) and n.value is not None:
                    tgt, val = n.target.id, n.value
                if tgt is None:
                    continue
                try:
                    consts[tgt] = ast.literal_eval(val)
                except Exception:
                    consts.pop(tgt, None)
        init = describe_class(classes[name][0], name)["methods"].get("__init__")
        out[name] = {"mro_in_package": order, "methods": sorted(m for m in methods if m != "__init__"), "properties": sorted(props),
                     "init_params": None if init is None else init["params"], "class_constants": consts}
    return out


def main():
    out = {
        "FluidEnv": describe_class(f"{REF}/envs/fluid_env.py", "FluidEnv"),
        "ParallelFluidEnv": describe_class(f"{REF}/envs/parallel_env.py", "ParallelFluidEnv"),
        "Simulation": describe_class(f"{REF}/simulation/simulation.py", "Simulation"),
        "EnvMode": describe_class(f"{REF}/types.py", "EnvMode"),
        "FluidEnvLike": describe_class(f"{REF}/types.py", "FluidEnvLike"),
        "registry": describe_functions(f"{REF}/registry.py"),
        "Config": describe_class(f"{REF}/config.py", "Config"),
        "package_all": ast.literal_eval(next(n.value for n in ast.parse(open(f"{REF}/__init__.py").read()).body
                                             if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "__all__")),
        "env_classes": env_class_surfaces(["CylinderJetEnv2D", "CylinderRotEnv2D", "CylinderJetEnv3D", "AirfoilEnv2D", "AirfoilEnv3D",
                                           "RBCEnv2D", "RBCEnv3D", "TCF3DBothEnv", "TCF3DBottomEnv"]),
    }
    with open(os.path.join(OUT, "reference_api.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out.items():
        if "methods" in v:
            print(k, len(v["methods"]), "methods,", len(v["properties"]), "properties")
        elif k == "env_classes":
            for c, d in v.items():
                print(" ", c, d["mro_in_package"], len(d["methods"]), "methods,", len(d["properties"]), "properties")
        elif isinstance(v, dict):
            print(k, sorted(v))
        else:
            print(k, v)


if __name__ == "__main__":
    main()
