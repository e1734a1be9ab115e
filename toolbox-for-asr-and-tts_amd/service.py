"""HTTP surface of the voice service's recognition route, over the GPU path of this package.

Mirrors, for the one route that reaches the hot path, what a client of the reference sees:

* ``POST /api/v1/voice/asr`` (/root/reference/voice-service/app/api/voice.py:52-89, mounted at
  ``/api/v1/voice`` by app/main.py:100) with the request / response models of app/models/voice.py:6-21 (same field
  names, types and defaults); payloads shorter than 100 characters are refused with HTTP 400 and any failure
  inside the route becomes HTTP 500, as there;
* ``GET /health`` -> ``{"status": "healthy", "service": "voice"}`` (app/main.py:191-193);
* ``WS /api/v1/voice/asr/ws`` (app/api/voice.py:92-910), the recognition part of its protocol with wake-word mode
  off: a welcome message on connect; every client message is JSON carrying a base64 WAV chunk in ``wav_base64`` or
  ``audio_data``; chunks go through ``StreamingPcmSession`` (the reference session's accumulation policy); when the
  session asks to finalize, ``{"type": "processing", "status": "finalizing"}`` is sent, the utterance held on the
  device runs through the engine and the decoder, and ``{"type": "result", "status": "completed", "text", "success"}``
  follows; malformed input is answered with ``{"type": "error", "message", "code"}`` using the reference's codes
  (EMPTY_MESSAGE, INVALID_JSON, MISSING_AUDIO_DATA, AUDIO_DECODE_ERROR, PROCESSING_ERROR) and the connection stays
  open.  The control messages are answered as there (voice.py:227-303): ``end_conversation`` and ``cancel_enrollment``
  reset the session and are confirmed with a ``status`` message, ``start_asr`` is only honoured in the enrolment
  modes, which this service never enters, so it is ignored without an answer as the reference does outside them.
  A decoder may return the reference session's sentinel strings (voice_interface.py:1922, 2187, 2217) and they are
  turned into the same client messages (voice.py:635-664).  Wake-word activation, speaker enrolment / verification
  and LLM correction themselves are not served.

What runs behind the route is this package: base64 WAV -> ``decode_wav_base64`` -> float32 samples in [-1, 1) ->
``AsrEngine`` (float PCM is scaled by 32768 in the fbank kernel, as FunASR's loader does with the WAV file the
reference hands it at voice_service.py:114: a 16-bit clip reaches the kernel with its original sample values) ->
encoder states.  Turning encoder states into text is the decoder's job, which is not part of this package: ``create_app``
takes it as a callable ``decoder(encoder_out [1, T', 512], lengths [1]) -> str``; without one the route answers
``success = false`` with a message saying so.  Wake-word, LLM post-correction and sample saving are accepted in the
request and ignored (they belong to the reference's control plane)."""
from __future__ import annotations

import asyncio
import contextlib
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Optional

import torch
from fastapi import FastAPI, HTTPException, WebSocket, WebSocketDisconnect
from fastapi.responses import JSONResponse
from pydantic import BaseModel

from .ingress import StreamingPcmSession, decode_wav_base64


class ASRRequest(BaseModel):
    audio_data: str
    use_wake: Optional[bool] = None
    use_llm: Optional[bool] = None
    save_sample: bool = False
    sample_id: Optional[str] = None
    diagnosis_session_id: str


class ASRResponse(BaseModel):
    text: str
    success: bool
    message: Optional[str] = None
    sample_id: Optional[str] = None


# What the reference's session returns instead of text when there is nothing to show, and what the route tells the client
# (voice-service/app/api/voice.py:635-664; the message texts are part of the wire protocol).
SENTINEL_MESSAGES = {
    "__SV_VERIFICATION_FAILED__": "抱歉，请再说一遍！",
    "__SV_NOT_ACTIVATED__": "非认证注册声音，拒绝访问。",
    "__ASR_RESULT_EMPTY__": "抱歉，请再说一遍！",
}
EMPTY_RESULT_MESSAGE = "抱歉，请再说一遍！"


def result_message(final_text: str) -> dict:
    """The ``result`` message for what finalize returned: text, a sentinel, or nothing."""
    if final_text and final_text not in SENTINEL_MESSAGES:
        return {"type": "result", "status": "completed", "text": final_text, "success": True}
    return {"type": "result", "status": "completed", "text": "", "success": False,
            "message": SENTINEL_MESSAGES.get(final_text, EMPTY_RESULT_MESSAGE)}


def recognize(engine, request: ASRRequest, decoder: Optional[Callable] = None) -> ASRResponse:
    """One clip through the engine (batch 1).  Clips longer than the engine's ``max_samples`` are refused."""
    samples, _ = decode_wav_base64(request.audio_data)
    n = int(samples.shape[0])
    if n > engine.max_samples:
        raise ValueError(f"clip of {n} samples exceeds the engine's {engine.max_samples}")
    pcm = torch.zeros((1, max(n, 8)), dtype=torch.float32)
    pcm[0, :n] = torch.from_numpy(samples)
    pcm = pcm.to(engine.device)
    lens = torch.tensor([n], dtype=torch.int32, device=engine.device)
    out, out_lens = engine.run(pcm, lens)
    if decoder is None:
        return ASRResponse(text="", success=False, sample_id=request.sample_id,
                           message=f"encoder states computed ({int(out_lens[0])} frames); no decoder is configured")
    text = decoder(out, out_lens)
    return ASRResponse(text=text, success=True, message=None, sample_id=request.sample_id)


def finalize_session(engine, session: StreamingPcmSession, decoder: Optional[Callable] = None) -> str:
    """What the reference session's ``finalize`` hands to its pipeline, run here: the utterance the session holds ->
    encoder states -> ``decoder`` -> text ("" without a decoder or without audio).  An engine built with
    ``session_float=True`` takes the float buffer as it is (the int16 hand-off happens in the kernel's load);
    any other engine gets the int16 hand-off itself.  Same bits either way."""
    n = len(session.audio)
    if n == 0 or decoder is None:
        return ""
    if n > engine.max_samples:
        raise ValueError(f"utterance of {n} samples exceeds the engine's {engine.max_samples}")
    if getattr(engine, "session_float", False):
        pcm = session.samples().reshape(1, -1)
        lens = torch.tensor([n], dtype=torch.int32, device=engine.device)
    else:
        pcm, lens = session.wire()
    out, out_lens = engine.run(pcm.to(engine.device), lens.to(engine.device))
    return decoder(out, out_lens) or ""


def head_decoder(head) -> Callable:
    """``decoder`` callable over a ``ParaformerHead`` (CIF predictor + decoder on the device): the utterance's text when
    the head knows the model's token list, else its token ids joined by spaces."""
    def decode(out, out_lens):
        res = head.recognize(out, out_lens)[0]
        return res if isinstance(res, str) else " ".join(str(t) for t in res)
    return decode


def create_app(engine, decoder: Optional[Callable] = None, clock: Optional[Callable[[], float]] = None, head=None,
               vad_model=None) -> FastAPI:
    """``clock``: the time source of the streaming sessions' silence timer (``time.time`` by default).
    ``engine``: an ``AsrEngine(batch=1, max_samples=...)`` (anything with ``run``, ``device`` and
    ``max_samples`` serves, which is how the CPU tests drive the route).  ``head``: a loaded ``ParaformerHead`` on the
    engine's context; it becomes the decoder when no ``decoder`` callable is given.  ``vad_model``: ``AutoModel`` of an
    FSMN-VAD directory; the streaming sessions then ask it once per chunk, as the reference's sessions do
    (voice_interface.py:1585-1590)."""
    if decoder is None and head is not None:
        decoder = head_decoder(head)
    app = FastAPI(title="voice")
    # The reference calls its models on the event-loop thread (voice.py:613, voice_interface.py:2049), which stalls
    # every other connection for the length of a recognition.  Here the engine runs on ONE worker thread: requests
    # queue up behind each other (an engine is a single batch slot and is not re-entrant) while the loop stays free.
    worker = ThreadPoolExecutor(max_workers=1, thread_name_prefix="asr-engine")

    async def on_engine(fn, *args):
        def call():
            dev = getattr(engine, "device", None)
            guard = torch.cuda.device(dev) if isinstance(dev, torch.device) and dev.type == "cuda" else contextlib.nullcontext()
            with guard:
                return fn(*args)
        return await asyncio.get_running_loop().run_in_executor(worker, call)

    @app.get("/health")
    async def health_check():
        return {"status": "healthy", "service": "voice"}

    @app.post("/api/v1/voice/asr", response_model=ASRResponse)
    async def speech_recognition(request: ASRRequest):
        try:
            if not request.audio_data or len(request.audio_data) < 100:
                raise HTTPException(status_code=400, detail="audio data missing or too short")
            result = await on_engine(recognize, engine, request, decoder)
            return JSONResponse(content=result.model_dump(), headers={"Content-Type": "application/json; charset=utf-8"})
        except HTTPException:
            raise
        except Exception as e:  # same mapping as the reference route: every failure inside is a 500
            raise HTTPException(status_code=500, detail=f"speech recognition failed: {e}") from e

    @app.websocket("/api/v1/voice/asr/ws")
    async def speech_recognition_websocket(websocket: WebSocket):
        import json
        import time
        now = clock or time.time
        await websocket.accept()
        try:
            session = StreamingPcmSession(device=engine.device, now=now(), vad_model=vad_model)
        except Exception as e:
            await websocket.send_json({"type": "error", "message": f"session creation failed: {e}",
                                       "code": "SESSION_CREATE_ERROR"})
            await websocket.close(code=1011, reason="Failed to create session")
            return
        await websocket.send_json({"type": "welcome", "message": "Connected to voice recognition service",
                                   "timestamp": now(), "use_wake": False, "mode": "ASR_ACTIVE"})
        while True:
            try:
                text = await websocket.receive_text()
                if not text:
                    await websocket.send_json({"type": "error", "message": "empty message ignored", "code": "EMPTY_MESSAGE"})
                    continue
                try:
                    data = json.loads(text)
                except json.JSONDecodeError:
                    await websocket.send_json({"type": "error", "message": "messages must be JSON", "code": "INVALID_JSON"})
                    continue
                kind = data.get("type") if isinstance(data, dict) else None
                if kind == "end_conversation":      # the user left: everything starts over (voice.py:227-237)
                    session.reset(now())
                    await websocket.send_json({"type": "status", "status": "conversation_ended",
                                               "message": "会话已结束，状态已重置"})
                    continue
                if kind == "cancel_enrollment":     # voice.py:240-264 (there the session also falls back to wake-word mode)
                    session.reset(now())
                    await websocket.send_json({"type": "status", "status": "enrollment_cancelled",
                                               "message": "声纹录制已取消，已回退到等待唤醒状态"})
                    continue
                if kind == "start_asr":             # only meaningful while enrolling (voice.py:267-302): ignored here
                    continue
                wav_b64 = (data.get("wav_base64") or data.get("audio_data")) if isinstance(data, dict) else None
                if not wav_b64 or not isinstance(wav_b64, str):
                    await websocket.send_json({"type": "error", "message": "field 'wav_base64' or 'audio_data' missing",
                                               "code": "MISSING_AUDIO_DATA"})
                    continue
                try:
                    chunk, _ = decode_wav_base64(wav_b64)
                except Exception as e:
                    await websocket.send_json({"type": "error", "message": f"audio decoding failed: {e}",
                                               "code": "AUDIO_DECODE_ERROR"})
                    continue
                try:
                    # on the engine's worker thread like every other device call: a chunk is copied into the session's
                    # device buffer and, with a VAD model, goes through its network; neither belongs on the event loop,
                    # and the one worker keeps device work in order
                    state = await on_engine(session.process_chunk, chunk, now())
                    if state["should_finalize"]:
                        await websocket.send_json({"type": "processing", "status": "finalizing",
                                                   "message": "processing audio"})
                        try:
                            final_text = await on_engine(finalize_session, engine, session, decoder)
                        finally:
                            # whatever happened to the utterance, the session starts over (the reference resets its
                            # recognition state after every finalize, voice.py:878): a failed utterance, e.g. one
                            # longer than the engine takes, must not be retried on every following silent chunk
                            session.reset(now())
                        await websocket.send_json(result_message(final_text))
                except Exception as e:
                    await websocket.send_json({"type": "error", "message": f"chunk processing failed: {e}",
                                               "code": "PROCESSING_ERROR"})
                    continue
            except WebSocketDisconnect:
                break

    return app
