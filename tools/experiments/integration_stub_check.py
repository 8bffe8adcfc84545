# run the INTEGRATION.md stub verbatim against the oracle-free torch expression
import re, sys, torch
sys.path.insert(0, ".")
src = open(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))), "INTEGRATION.md")).read()
code = re.search(r"```python\nimport ctypes, torch\n(.*?)```", src, re.S).group(0)[len("```python\n"):-3]
ns = {}
exec(code, ns)
x = torch.randn(300, 64, device="cuda"); w = torch.randn(70, 64, device="cuda")
idx = ns["nearest_codes"](x, w)
d = (x.double() ** 2).sum(1, keepdim=True) + (w.double() ** 2).sum(1) - 2 * x.double() @ w.double().t()
print("stub indices equal fp64 argmin on", int((idx == d.argmin(1)).sum()), "of 300")
