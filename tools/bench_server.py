"""The whole serving stack on one GPU, measured the way the reference measures itself (load
generator -> HTTP -> scheduler -> prefill / graph decode -> event stream): a Llama-3-8B-shaped
random-weight engine behind uvicorn on a loopback socket, N concurrent streaming requests.
   python tools/bench_server.py [N=32] [PROMPT=512] [GEN=128] [MAX_BATCH=N]
MAX_BATCH below N makes requests queue behind the running batch (continuous batching).
Sampled decoding (the load generator's temperatures), one token per character of the prompt."""
import os, socket, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvicorn
import micro_vllm_amd  # noqa: F401
from micro_vllm_amd.api_server import EngineLoop, create_app
from micro_vllm_amd.core.engine import InferenceEngine
from micro_vllm_amd.models import SHAPES, build_synthetic_model

n, plen, gen = (int(a) for a in (sys.argv[1:4] + ["32", "512", "128"][len(sys.argv[1:4]):]))
max_bs = int(sys.argv[4]) if len(sys.argv) > 4 else max(n, 1)


class Tok:
    eos_token_id = 1

    def __init__(self, vocab):
        self.vocab = vocab

    def encode(self, prompt, add_special_tokens=True):
        return [5 + (ord(c) * 7919) % (self.vocab - 5) for c in prompt]

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(i) for i in ids)


cfg = SHAPES["llama3-8b"]
model = build_synthetic_model(cfg, "cuda", seed=1234)
blocks = n * ((plen + gen + 255) // 256 + 1) + 4
eng = InferenceEngine(model=model, tokenizer=Tok(cfg.vocab_size), max_batch_size=max_bs,
                      max_prefill_tokens=max(2048, plen), n_blocks=blocks, max_tokens=plen + gen + 256,
                      graph_batch_sizes=[b for b in (1, 2, 4, 8, 16, 32, 40) if b <= max_bs] or [max_bs])
loop = EngineLoop(eng)
loop.start()
sock = socket.socket()
sock.bind(("127.0.0.1", 0))
port = sock.getsockname()[1]
server = uvicorn.Server(uvicorn.Config(create_app(loop), log_level="warning"))
th = threading.Thread(target=lambda: server.run(sockets=[sock]), daemon=True)
th.start()
while not server.started:
    time.sleep(0.05)

try:
    for rep in range(2):
        s0 = loop.steps
        # the load generator is a process of its own, as in a deployment: inside this one it would
        # take the interpreter lock from the engine thread
        r = subprocess.run([sys.executable, "-m", "micro_vllm_amd.benchmark", str(n), "--url",
                            f"http://127.0.0.1:{port}", "--max-tokens", str(gen), "--seed", str(rep),
                            "--prompt-chars", str(plen)], capture_output=True, text=True, timeout=240,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        print(f"rep {rep}: {r.stdout.strip() or r.stderr[-2000:]}  engine steps {loop.steps - s0}", flush=True)
finally:
    server.should_exit = True
    th.join(timeout=10)
    loop.stop()
