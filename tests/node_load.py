"""TESTS ONLY: load generator for tests/test_node_scale.py.  Runs in its OWN processes (a Python HTTP client costs more CPU per
request than the routing process it is measuring: building a 43 KB query string in httpx takes 4 ms), one thread and one
keep-alive `http.client` connection per session, request lines pre-built."""
import http.client
import json
import threading
import time


def _post(conn, path):
    conn.request("POST", path, headers={"Content-Length": "0"})
    r = conn.getresponse()
    body = r.read()
    return r.status, body


def drive(host, port, sids, rounds, chunk_b64, start_at):
    """Every session: `rounds` x (add_chunk, process).  Returns (per-process-call latencies, errors, t_first, t_last)."""
    lat, errs = [], []
    lock = threading.Lock()

    def session(sid):
        conn = http.client.HTTPConnection(host, port, timeout=120)
        add = f"/session/{sid}/add_chunk?audio_data={chunk_b64}"
        proc = f"/session/{sid}/process"
        try:
            conn.connect()
            time.sleep(max(0.0, start_at - time.time()))
            for _ in range(rounds):
                st, body = _post(conn, add)
                if st != 200:
                    raise RuntimeError(f"add_chunk {st} {body[:200]!r}")
                t0 = time.monotonic()
                st, body = _post(conn, proc)
                dt = time.monotonic() - t0
                if st != 200 or len(json.loads(body)["uncommited_words"]) != 1:
                    raise RuntimeError(f"process {st} {body[:200]!r}")
                with lock:
                    lat.append(dt)
        except Exception as e:  # noqa: BLE001
            with lock:
                errs.append(repr(e))
        finally:
            conn.close()

    th = [threading.Thread(target=session, args=(s,)) for s in sids]
    [t.start() for t in th]
    [t.join() for t in th]
    return lat, errs, time.time()
