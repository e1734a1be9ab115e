"""HTTP surface of the tts-service: the job routes a client of the reference polls, over a synthesis hook.

Mirrors what a client of the reference sees:

* ``POST /api/v1/tts/start``  ``{text, voice = "zhitian_emo"}`` -> ``{status: "started", job_id, message}``; empty or
  blank text is refused with HTTP 400, any failure while creating the job is HTTP 500
  (/root/reference/tts-service/app/api/tts.py:27-67, models app/models/tts.py:7-33);
* ``POST /api/v1/tts/cancel`` ``{job_id}`` -> ``{status: cancelled | already_completed | already_cancelled, ...}``,
  404 for an unknown job (tts.py:70-82, tts_service.py:998-1012);
* ``GET /api/v1/tts/result/{job_id}`` -> ``{status: processing | completed | cancelled | error, job_id, audio_base64,
  text, audio_size, message, error}``; an ``error`` job is a normal JSON answer, not an HTTP error (tts.py:85-119,
  tts_service.py:1014-1029);
* ``DELETE /api/v1/tts/jobs/{job_id}`` -> ``{status: "deleted", job_id}``; 404 unknown, 400 while the job is still
  running (tts.py:122-133, tts_service.py:1031-1041);
* ``GET /health`` -> ``{status: "healthy", model_loaded, active_jobs}`` (app/main.py:158-173);
  the routes are mounted at ``/api/v1/tts`` (app/main.py:156).

The job table is in memory and jobs run on a pool of two worker threads, as in the reference
(tts_service.py:47,57,964-986).  What turns text into a waveform (Sambert acoustic model + HiFi-GAN vocoder in the
reference, tts_service.py:617-625) is NOT part of this module: ``create_tts_app`` takes it as
``synthesize(text, voice) -> bytes`` (a 16-bit mono WAV file image, tts_service.py:396-411); without one every job
ends in ``status = "error"`` with a message saying so.  A job cancelled while it runs keeps ``cancelled`` (the worker
checks before it stores its result, tts_service.py:590,707).
"""
from __future__ import annotations

import asyncio
import base64
import time
import uuid
from concurrent.futures import ThreadPoolExecutor
from datetime import datetime
from typing import Callable, Dict, Optional

from fastapi import FastAPI, HTTPException
from pydantic import BaseModel, Field


class TTSRequest(BaseModel):
    text: str = Field(..., min_length=1)
    voice: str = Field(default="zhitian_emo")


class CancelRequest(BaseModel):
    job_id: str = Field(...)


class TTSResponse(BaseModel):
    status: str = Field(...)
    job_id: str = Field(...)
    message: str = Field(default="")


class TTSResultResponse(BaseModel):
    status: str = Field(...)
    job_id: str = Field(...)
    audio_base64: str = Field(default="")
    text: str = Field(default="")
    audio_size: int = Field(default=0)
    message: str = Field(default="")
    error: str = Field(default="")


class TTSJobs:
    """The reference manager's job table: start / cancel / result / cleanup with the same status words."""

    def __init__(self, synthesize: Optional[Callable[[str, str], bytes]] = None, workers: int = 2):
        self.synthesize = synthesize
        self.jobs: Dict[str, dict] = {}
        self._executor = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="tts")

    @property
    def model_loaded(self) -> bool:
        return self.synthesize is not None

    def active_jobs(self) -> int:
        return len([j for j in self.jobs.values() if j["status"] in ("pending", "processing")])

    def _run(self, job_id: str, text: str, voice: str):
        job = self.jobs.get(job_id)
        if job is None or job["status"] == "cancelled":
            return
        job["status"] = "processing"
        try:
            if self.synthesize is None:
                raise RuntimeError("no synthesis backend is configured")
            wav = self.synthesize(text, voice)
            if self.jobs.get(job_id, {}).get("status") == "cancelled":
                return
            job["result"] = {"audio_base64": base64.b64encode(wav).decode("ascii"), "text": text, "audio_size": len(wav)}
            job["_elapsed_time_ms"] = 1e3 * (time.time() - job["start_time"])
            job["status"] = "completed"
        except Exception as e:  # the reference stores the text of any failure in the job (tts_service.py:950-960)
            if job_id in self.jobs and self.jobs[job_id]["status"] != "cancelled":
                job["status"] = "error"
                job["error"] = str(e)

    async def start_task(self, text: str, voice: str = "zhitian_emo") -> str:
        job_id = str(uuid.uuid4())
        self.jobs[job_id] = {"status": "pending", "text": text, "voice": voice,
                             "created_at": datetime.now().isoformat(), "start_time": time.time()}
        asyncio.get_running_loop().run_in_executor(self._executor, self._run, job_id, text, voice)
        return job_id

    async def cancel_task(self, job_id: str) -> dict:
        if job_id not in self.jobs:
            return {"status": "not_found", "message": f"任务 {job_id} 不存在"}
        job = self.jobs[job_id]
        if job["status"] == "completed":
            return {"status": "already_completed", "message": "任务已完成，无法取消"}
        if job["status"] == "cancelled":
            return {"status": "already_cancelled", "message": "任务已被取消"}
        job["status"] = "cancelled"
        job["cancelled_at"] = datetime.now().isoformat()
        return {"status": "cancelled", "message": "任务已取消"}

    async def get_result(self, job_id: str) -> dict:
        if job_id not in self.jobs:
            return {"status": "not_found", "job_id": job_id}
        job = self.jobs[job_id]
        st = job["status"]
        if st in ("pending", "processing"):
            return {"status": "processing", "job_id": job_id, "message": "任务处理中，请稍后重试"}
        if st == "cancelled":
            return {"status": "cancelled", "job_id": job_id, "message": "任务已取消"}
        if st == "error":
            return {"status": "error", "job_id": job_id, "error": job.get("error", "未知错误")}
        if st == "completed":
            return {"status": "completed", "job_id": job_id, **job["result"]}
        return {"status": "unknown", "job_id": job_id, "message": "未知的任务状态"}

    async def cleanup_job(self, job_id: str) -> dict:
        if job_id not in self.jobs:
            return {"status": "not_found", "message": f"任务 {job_id} 不存在"}
        if self.jobs[job_id]["status"] not in ("completed", "cancelled", "error"):
            return {"status": "cannot_cleanup", "message": "只能清理已完成/已取消/失败的任务"}
        del self.jobs[job_id]
        return {"status": "deleted"}


def create_tts_app(synthesize: Optional[Callable[[str, str], bytes]] = None) -> FastAPI:
    app = FastAPI(title="tts")
    manager = TTSJobs(synthesize)
    app.state.tts_jobs = manager

    @app.post("/api/v1/tts/start", response_model=TTSResponse)
    async def start_tts(request: TTSRequest):
        try:
            if not request.text or not request.text.strip():
                raise HTTPException(status_code=400, detail="text 不能为空")
            job_id = await manager.start_task(request.text, request.voice)
            return TTSResponse(status="started", job_id=job_id, message="TTS 任务已启动")
        except HTTPException:
            raise
        except Exception as e:
            raise HTTPException(status_code=500, detail=f"启动 TTS 任务失败: {e}") from e

    @app.post("/api/v1/tts/cancel", response_model=TTSResponse)
    async def cancel_tts(request: CancelRequest):
        result = await manager.cancel_task(request.job_id)
        if result["status"] == "not_found":
            raise HTTPException(status_code=404, detail=f"任务 {request.job_id} 不存在")
        return TTSResponse(status=result["status"], job_id=request.job_id, message=result["message"])

    @app.get("/api/v1/tts/result/{job_id}", response_model=TTSResultResponse)
    async def get_tts_result(job_id: str):
        result = await manager.get_result(job_id)
        if result["status"] == "not_found":
            raise HTTPException(status_code=404, detail=f"任务 {job_id} 不存在")
        return TTSResultResponse(**result)

    @app.delete("/api/v1/tts/jobs/{job_id}")
    async def cleanup_job(job_id: str):
        result = await manager.cleanup_job(job_id)
        if result["status"] == "not_found":
            raise HTTPException(status_code=404, detail=f"任务 {job_id} 不存在")
        if result["status"] == "cannot_cleanup":
            raise HTTPException(status_code=400, detail=result["message"])
        return {"status": "deleted", "job_id": job_id}

    @app.get("/health")
    async def health():
        return {"status": "healthy", "model_loaded": manager.model_loaded, "active_jobs": manager.active_jobs()}

    return app
