"""Caller-contract fixture of seam S2 (SURVEY 8c "captured structural trace"): every attribute the reference's own callers
read or call on the scene model object.  Run in the BUILD container (it reads /root/reference; nothing of the reference
travels - the output is a list of names and call shapes):

    python tests/golden/gen_caller_contract.py   ->  tests/golden/caller_contract.json

For each of flow3d/trainer.py, flow3d/validator.py, flow3d/renderer.py the script walks the AST, collects every
`<expr>.model.<attr>...` / `model.<attr>...` access (the whole dotted chain, e.g. `fg.densify_params`,
`move_model.RT_main.parameters`), and for calls the positional-argument count and keyword names."""
import ast
import json
import os
import sys

REF = os.environ.get("D4GS_REFERENCE", "/root/reference")
FILES = ["flow3d/trainer.py", "flow3d/validator.py", "flow3d/renderer.py"]


def _is_model(node):
    return (isinstance(node, ast.Attribute) and node.attr == "model") or (isinstance(node, ast.Name) and node.id == "model")


def _chain(node):
    """dotted attribute names hanging off the model object, outermost last; None if `node` is not rooted at it"""
    names = []
    while isinstance(node, ast.Attribute) and not _is_model(node):
        names.append(node.attr)
        node = node.value
    return names[::-1] if _is_model(node) and names else None


def collect(path):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    calls = {id(c.func): c for c in ast.walk(tree) if isinstance(c, ast.Call)}
    inner = {id(n.value) for n in ast.walk(tree) if isinstance(n, ast.Attribute)}  # not the end of its chain
    out = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Attribute) and id(n) not in inner:
            ch = _chain(n)
            if ch is None:
                continue
            rec = {"file": path, "line": n.lineno, "attr": ch[0], "chain": ".".join(ch), "call": id(n) in calls}
            if rec["call"]:
                c = calls[id(n)]
                rec["n_pos"] = len(c.args)
                rec["kwargs"] = sorted(k.arg for k in c.keywords if k.arg)
            out.append(rec)
    return out


if __name__ == "__main__":
    recs = [r for f in FILES for r in collect(f)]
    recs.sort(key=lambda r: (r["chain"], r["file"], r["line"]))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "caller_contract.json")
    json.dump(recs, open(dst, "w"), indent=1)
    print(f"{len(recs)} accesses, {len({r['attr'] for r in recs})} distinct attributes -> {dst}", file=sys.stderr)
