import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, micro_vllm_amd
from micro_vllm_amd.core.layer.sampler import sample, greedy, Sampler
dev="cuda"
for bs in (1, 32):
    logits=(torch.randn(bs,128256,device=dev)*3).to(torch.bfloat16)
    t=torch.full((bs,),0.7,device=dev); p=torch.full((bs,),0.9,device=dev); u=torch.rand(bs,device=dev)
    def tm(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)/reps*1e3
    print(f"bs={bs}: device sample k=50 {tm(lambda: sample(logits,t,p,50,uniforms=u)):.1f} us, k=1000 {tm(lambda: sample(logits,t,p,1000,uniforms=u)):.1f} us, argmax {tm(lambda: greedy(logits)):.1f} us, torch recipe k=50 {tm(lambda: Sampler._sample_impl(logits,t,p,50)):.1f} us")
