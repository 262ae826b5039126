import sys, json
d = json.loads(sys.stdin.read())
print(round(d['value'], 1), {k: round(v['ms'] * 1e3, 1) for k, v in d['kernels'].items()})
