"""CPU: the three command lines keep the reference's flags (pretrain_e4t.py:66-122, tuning_e4t.py:26-63, inference.py:34-50);
CUDA-only switches are refused loudly instead of being ignored."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(module, argv, monkeypatch):
    monkeypatch.setattr(sys, "argv", [f"{module}.py"] + argv)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module(module).parse_args()


def test_pretrain_flags(monkeypatch):
    a = parse("pretrain_e4t", ["--pretrained_model_name_or_path", "sd", "--clip_model_name_or_path", "ViT-H-14::laion2b_s32b_b79k",
                               "--domain_class_token", "art", "--placeholder_token", "*s", "--prompt_template", "art", "--reg_lambda", "0.01",
                               "--train_image_dataset", "imgs", "--webdataset", "--resolution", "512", "--train_batch_size", "16", "--learning_rate", "1e-6",
                               "--scale_lr", "--lr_scheduler", "cosine", "--lr_warmup_steps", "100", "--gradient_accumulation_steps", "2",
                               "--max_train_steps", "30000", "--dataloader_num_workers", "8", "--output_dir", "o", "--seed", "1", "--mixed_precision", "bf16",
                               "--enable_xformers_memory_efficient_attention", "--checkpointing_steps", "1000", "--log_steps", "500",
                               "--save_sample_prompt", "a photo of *s", "--n_save_sample", "2", "--save_guidance_scale", "7.5", "--save_inference_steps", "20",
                               "--resume_from_checkpoint", "latest", "--unfreeze_clip_vision", "--domain_embed_scale", "0.1"], monkeypatch)
    assert a.webdataset and a.gradient_accumulation_steps == 2 and a.lr_scheduler == "cosine" and a.resume_from_checkpoint == "latest"
    for bad in (["--use_8bit_adam"], ["--mixed_precision", "fp16"], ["--gradient_accumulation_steps", "0"], ["--lr_scheduler", "exponential"]):
        with pytest.raises(SystemExit):
            parse("pretrain_e4t", bad, monkeypatch)
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert parse("pretrain_e4t", ["--synthetic_data"], monkeypatch).local_rank == 3


def test_tuning_flags(monkeypatch):
    a = parse("tuning_e4t", ["--pretrained_model_name_or_path", "ckpt", "--train_image_path", "x.png", "--reg_lambda", "0.1", "--max_train_steps", "30",
                             "--gradient_accumulation_steps", "2", "--lr_scheduler", "constant_with_warmup", "--lr_warmup_steps", "3", "--unfreeze_clip_vision",
                             "--scale_lr", "--checkpointing_steps", "10", "--max_grad_norm", "1.0"], monkeypatch)
    assert a.learning_rate == 1.6e-5 and a.seed == 42 and a.train_batch_size == 16 and a.max_grad_norm == 1.0      # the reference's defaults
    assert parse("tuning_e4t", ["--train_text_encoder"], monkeypatch).train_text_encoder          # native since round 2 (e4t/text.py)
    assert parse("tuning_e4t", [], monkeypatch).prompt_template is None                           # None -> the pre-trained run's template (:252-253)
    for bad in (["--use_8bit_adam"], ["--gradient_accumulation_steps", "0"]):
        with pytest.raises(SystemExit):
            parse("tuning_e4t", bad, monkeypatch)


def test_inference_flags(monkeypatch):
    a = parse("inference", ["--pretrained_model_name_or_path", "ckpt", "--image_path_or_url", "x.png", "--prompt", "a photo of *s::a painting of *s",
                            "--num_inference_steps", "30", "--guidance_scale", "7.5", "--num_images_per_prompt", "2", "--scheduler_type", "dpm_solver++",
                            "--seed", "3", "--height", "512", "--width", "512"], monkeypatch)
    assert a.scheduler_type == "dpm_solver++" and a.guidance_scale == 7.5
    d = parse("inference", [], monkeypatch)
    assert (d.scheduler_type, d.num_inference_steps, d.guidance_scale, d.prompt) == ("ddim", 50, 1.0, "a photo of *s")   # inference.py:40-46
    with pytest.raises(SystemExit):
        parse("inference", ["--scheduler_type", "heun"], monkeypatch)
