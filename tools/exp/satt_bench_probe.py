import sys, json
sys.path.insert(0,__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))); sys.path.insert(0, sys.path[0] + '/image-captioning_amd')
import torch, bench
out = bench.variants_bench(torch.device("cuda",0))
print(json.dumps({k:out[k] for k in out if 'softattn' in k}, indent=1))
