"""Load generator for the HTTP front end: N concurrent streaming requests against
/generate_stream, as the reference measures itself (benchmark.py:7-95: one aiohttp session, all
requests at once, one server-sent event per generated token).

Reported: generated tokens, wall-clock time of the whole job and tokens per wall-clock second
(the serving throughput), the reference's own figure (tokens divided by the SUM of the request
durations, benchmark.py:89-91, kept for comparability), and time to first token / per-request
latency percentiles, which the reference does not measure.

    python -m micro_vllm_amd.benchmark 32 --url http://127.0.0.1:8000 --max-tokens 500
"""
import argparse
import asyncio
import json
import random
import time

TOPICS = ["a file upload handler", "consensus in distributed ledgers", "a rate-limited task pool",
          "remote work", "learning network security", "SQL and NoSQL stores", "a linked list",
          "monoliths and microservices", "type-safe state management", "containers and virtual machines",
          "overfitting and underfitting", "a small neural network", "REST and GraphQL", "smart pointers",
          "reading technical documentation", "a tiny RPC framework", "attention in language models"]
FORMS = ["Explain {} to a new engineer.", "Write a short essay about {}.", "Compare the trade-offs of {}.",
         "Give a step-by-step study plan for {}.", "Write example code for {}."]


def make_requests(n: int, max_tokens: int, seed: int = 0, prompt_chars: int = 0):
    """`prompt_chars` > 0 pads / cuts every prompt to that many characters (prompts of a known
    length for a server whose tokenizer maps characters to tokens)."""
    rng = random.Random(seed)
    reqs = [{"prompt": rng.choice(FORMS).format(rng.choice(TOPICS)) + f" (request {i})",
             "max_tokens": max_tokens, "temperature": rng.choice([0.5, 0.6, 0.7, 0.8, 0.9]),
             "top_p": 0.9, "stream": True} for i in range(n)]
    if prompt_chars > 0:
        for i, r in enumerate(reqs):
            fill = "".join(chr(33 + (i * 31 + j * 7) % 90) for j in range(prompt_chars))
            r["prompt"] = (r["prompt"] + " " + fill)[:prompt_chars]
    return reqs


async def send_request(session, url: str, data: dict):
    t0 = time.time()
    first, tokens, text = None, 0, ""
    async with session.post(url + "/generate_stream", json=data) as resp:
        resp.raise_for_status()
        async for line in resp.content:
            line = line.decode("utf-8").strip()
            if not line.startswith("data:"):
                continue
            event = json.loads(line[5:])
            if "error" in event:
                raise RuntimeError(event["error"])
            tokens += 1
            if first is None:
                first = time.time() - t0
            text = event.get("full_text", text)
    return {"tokens": tokens, "seconds": time.time() - t0, "ttft": first, "text": text}


def summarize(results, wall: float) -> dict:
    tokens = sum(r["tokens"] for r in results)
    total = sum(r["seconds"] for r in results)
    lat = sorted(r["seconds"] for r in results)
    ttft = sorted(r["ttft"] for r in results if r["ttft"] is not None)

    def pct(xs, q):
        return xs[min(len(xs) - 1, int(q * len(xs)))] if xs else None

    return {"requests": len(results), "tokens": tokens, "wall_s": round(wall, 4),
            "tokens_per_s": round(tokens / wall, 2) if wall > 0 else 0.0,
            "tokens_per_summed_request_s": round(tokens / total, 2) if total > 0 else 0.0,
            "ttft_s_median": pct(ttft, 0.5), "ttft_s_p99": pct(ttft, 0.99),
            "latency_s_median": pct(lat, 0.5), "latency_s_p99": pct(lat, 0.99)}


async def run(url: str, n: int, max_tokens: int, seed: int = 0, session=None, prompt_chars: int = 0) -> dict:
    import aiohttp
    own = session is None
    session = session or aiohttp.ClientSession(timeout=aiohttp.ClientTimeout(total=None))
    try:
        t0 = time.time()
        results = await asyncio.gather(*[send_request(session, url, d) for d in make_requests(n, max_tokens, seed, prompt_chars)])
        return summarize(results, time.time() - t0)
    finally:
        if own:
            await session.close()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("batch_size", type=int, nargs="?", default=32)
    ap.add_argument("--url", default="http://127.0.0.1:8000")
    ap.add_argument("--max-tokens", type=int, default=500)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--prompt-chars", type=int, default=0, help="pad / cut every prompt to this many characters")
    args = ap.parse_args(argv)
    print(json.dumps(asyncio.run(run(args.url, args.batch_size, args.max_tokens, args.seed, prompt_chars=args.prompt_chars))))


if __name__ == "__main__":
    main()
