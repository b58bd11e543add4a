"""The command-line server as a user would start it: writes a synthetic checkpoint directory, starts
`python -m micro_vllm_amd.api_server DIR` as a child process, waits for /health, drives it with the
load generator, stops it.
   python tools/serve_check.py [SHAPE=tiny-d128] [REQUESTS=8] [GEN=32]"""
import asyncio, json, os, socket, subprocess, sys, tempfile, time, urllib.request
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from micro_vllm_amd import benchmark

shape = sys.argv[1] if len(sys.argv) > 1 else "tiny-d128"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
gen = int(sys.argv[3]) if len(sys.argv) > 3 else 32
work = tempfile.mkdtemp(prefix="mv_ckpt_")
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_checkpoint.py"), shape, work],
               check=True, cwd=ROOT)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
server = subprocess.Popen([sys.executable, "-m", "micro_vllm_amd.api_server", work, "--host", "127.0.0.1",
                           "--port", str(port), "--max-batch-size", "8", "--max-tokens", "1024",
                           "--kv-blocks", "40", "--exact-rope"], cwd=ROOT)
try:
    url = f"http://127.0.0.1:{port}"
    for _ in range(600):
        if server.poll() is not None:
            raise SystemExit(f"server exited with {server.returncode}")
        try:
            health = json.load(urllib.request.urlopen(url + "/health", timeout=1))
            break
        except Exception:
            time.sleep(0.25)
    else:
        raise SystemExit("server did not come up")
    print("health:", health, flush=True)
    req = urllib.request.Request(url + "/generate", json.dumps({"prompt": "Hello", "max_tokens": 8,
                                                                "temperature": 0.0}).encode(),
                                 {"Content-Type": "application/json"})
    print("generate:", json.load(urllib.request.urlopen(req, timeout=60)), flush=True)
    print("load:", asyncio.run(benchmark.run(url, n, gen)), flush=True)
finally:
    server.terminate()
    try:
        server.wait(timeout=20)
    except subprocess.TimeoutExpired:
        server.kill()
