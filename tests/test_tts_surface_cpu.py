"""The tts-service job routes (start / cancel / result / cleanup / health) with a stand-in synthesis hook: status
words, HTTP codes and payload fields as the reference's clients see them.  The field tables are data in
tests/golden/tts_schema.json."""
import base64
import importlib
import json
import os
import threading
import time

import pytest

tts = importlib.import_module("toolbox_for_asr_and_tts_amd.tts_surface")
from fastapi.testclient import TestClient  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "tts_schema.json")


def _fields(model):
    sch = model.model_json_schema()
    return {name: [p["type"], name in sch.get("required", []), p.get("default")] for name, p in sch["properties"].items()}


def test_models_have_the_reference_fields():
    g = json.load(open(GOLDEN))
    for name in ("TTSRequest", "CancelRequest", "TTSResponse", "TTSResultResponse"):
        assert _fields(getattr(tts, name)) == g[name], name


def _poll(client, job_id, want, timeout=10.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        r = client.get(f"/api/v1/tts/result/{job_id}")
        assert r.status_code == 200
        if r.json()["status"] in want:
            return r.json()
        time.sleep(0.01)
    raise AssertionError(f"job never reached {want}: {r.json()}")


@pytest.mark.timeout(60)
def test_job_life_cycle():
    g = json.load(open(GOLDEN))
    gate = threading.Event()

    def synth(text, voice):
        if text == "slow":
            gate.wait(10)
        if text == "boom":
            raise ValueError("synthesis failed on purpose")
        return b"RIFF" + f"{voice}:{text}".encode()

    with TestClient(tts.create_tts_app(synth)) as client:
        h = client.get("/health").json()
        assert sorted(h) == sorted(g["health_keys"]) and h == {"status": "healthy", "model_loaded": True, "active_jobs": 0}
        # start -> poll -> completed
        r = client.post("/api/v1/tts/start", json={"text": "你好"})
        assert r.status_code == 200 and r.json()["status"] == "started" and set(r.json()) == set(g["TTSResponse"])
        job = r.json()["job_id"]
        done = _poll(client, job, {"completed"})
        assert set(done) == set(g["TTSResultResponse"])
        wav = base64.b64decode(done["audio_base64"])
        assert wav == b"RIFF" + "zhitian_emo:你好".encode() and done["audio_size"] == len(wav) and done["text"] == "你好"
        # refusals
        assert client.post("/api/v1/tts/start", json={"text": "   "}).status_code == 400
        assert client.post("/api/v1/tts/start", json={"text": ""}).status_code == 422
        assert client.post("/api/v1/tts/start", json={}).status_code == 422
        assert client.get("/api/v1/tts/result/nope").status_code == 404
        assert client.post("/api/v1/tts/cancel", json={"job_id": "nope"}).status_code == 404
        assert client.delete("/api/v1/tts/jobs/nope").status_code == 404
        # a finished job cannot be cancelled, but can be cleaned up
        assert client.post("/api/v1/tts/cancel", json={"job_id": job}).json()["status"] == "already_completed"
        assert client.delete(f"/api/v1/tts/jobs/{job}").json() == {"status": "deleted", "job_id": job}
        assert client.get(f"/api/v1/tts/result/{job}").status_code == 404
        # a failing job is an ordinary JSON answer with status "error"
        bad = client.post("/api/v1/tts/start", json={"text": "boom", "voice": "v2"}).json()["job_id"]
        e = _poll(client, bad, {"error"})
        assert e["error"] == "synthesis failed on purpose" and e["audio_base64"] == ""
        # a running job: processing, counted as active, not removable; cancelled while running it stays cancelled
        slow = client.post("/api/v1/tts/start", json={"text": "slow"}).json()["job_id"]
        assert client.get(f"/api/v1/tts/result/{slow}").json()["status"] == "processing"
        assert client.get("/health").json()["active_jobs"] == 1
        assert client.delete(f"/api/v1/tts/jobs/{slow}").status_code == 400
        assert client.post("/api/v1/tts/cancel", json={"job_id": slow}).json()["status"] == "cancelled"
        assert client.post("/api/v1/tts/cancel", json={"job_id": slow}).json()["status"] == "already_cancelled"
        gate.set()
        time.sleep(0.2)
        assert client.get(f"/api/v1/tts/result/{slow}").json()["status"] == "cancelled"
        assert client.delete(f"/api/v1/tts/jobs/{slow}").status_code == 200


@pytest.mark.timeout(60)
def test_without_a_backend_jobs_fail_loudly():
    with TestClient(tts.create_tts_app(None)) as client:
        assert client.get("/health").json()["model_loaded"] is False
        job = client.post("/api/v1/tts/start", json={"text": "hello"}).json()["job_id"]
        e = _poll(client, job, {"error"})
        assert "no synthesis backend" in e["error"]
