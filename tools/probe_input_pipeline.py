import sys, json, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/imbalanced-regression_amd')
import bench
print(json.dumps(bench.input_pipeline_probe(torch.device('cuda'), 10193.0), indent=1))
