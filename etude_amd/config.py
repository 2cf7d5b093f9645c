"""Configuration objects with the reference's field names and defaults.

``AMTAPC_Extractor`` accepts either the reference's own pydantic ``ExtractorConfig``
(etude/config/schema.py:123-131) or this dataclass mirror -- only attribute access is used, so the
two are interchangeable.  Defaults are copied from etude/config/schema.py:68-121 and :204-226.
"""
from __future__ import annotations

from dataclasses import dataclass, field


@dataclass
class ExtractorFeatureConfig:      # schema.py:68-79
    sr: int = 16000
    hop_sample: int = 256
    mel_bins: int = 256
    n_bins: int = 256
    fft_bins: int = 2048
    window_length: int = 2048
    log_offset: float = 1e-8
    window: str = "hann"
    pad_mode: str = "constant"     # NOT passed by _wav2feature (extractor.py:186-193): torchaudio default "reflect" applies


@dataclass
class ExtractorInputConfig:        # schema.py:82-88
    margin_b: int = 32
    margin_f: int = 32
    num_frame: int = 512
    min_value: float = -18.0


@dataclass
class ExtractorMidiConfig:         # schema.py:91-97
    note_min: int = 21
    note_max: int = 108
    num_note: int = 88
    num_velocity: int = 128


@dataclass
class ExtractorModelConfig:        # schema.py:100-112
    cnn_channel: int = 4
    cnn_kernel: int = 5
    dropout: float = 0.1
    transformer_hid_dim: int = 256
    transformer_pf_dim: int = 512
    encoder_n_head: int = 4
    encoder_n_layer: int = 3
    decoder_n_head: int = 4
    decoder_n_layer: int = 3
    sv_dim: int = 24


@dataclass
class ExtractorInferConfig:        # schema.py:115-121
    onset_threshold: float = 0.5
    offset_threshold: float = 1.0
    frame_threshold: float = 0.5
    min_duration: float = 0.08


@dataclass
class ExtractorConfig:             # schema.py:123-131
    feature: ExtractorFeatureConfig = field(default_factory=ExtractorFeatureConfig)
    input: ExtractorInputConfig = field(default_factory=ExtractorInputConfig)
    midi: ExtractorMidiConfig = field(default_factory=ExtractorMidiConfig)
    model: ExtractorModelConfig = field(default_factory=ExtractorModelConfig)
    infer: ExtractorInferConfig = field(default_factory=ExtractorInferConfig)


@dataclass
class HFTFeatureConfig:            # schema.py:161-172
    sr: int = 16000
    hop_sample: int = 256
    mel_bins: int = 256
    n_bins: int = 256
    fft_bins: int = 2048
    window_length: int = 2048
    log_offset: float = 1e-8
    window: str = "hann"
    pad_mode: str = "constant"     # IS passed to MelSpectrogram here (hft_transformer.py:130)


@dataclass
class HFTInputConfig:              # schema.py:175-181
    margin_b: int = 32
    margin_f: int = 32
    num_frame: int = 128
    min_value: float = -80.0


@dataclass
class HFTInferConfig:              # schema.py:184-192
    mode: str = "combination"
    thred_mpe: float = 0.5
    thred_onset: float = 0.75
    thred_offset: float = 0.5
    n_stride: int = 32
    bpm: float = 120.0


@dataclass
class HFTConfig:                   # schema.py:195-201
    feature: HFTFeatureConfig = field(default_factory=HFTFeatureConfig)
    input: HFTInputConfig = field(default_factory=HFTInputConfig)
    midi: ExtractorMidiConfig = field(default_factory=ExtractorMidiConfig)
    infer: HFTInferConfig = field(default_factory=HFTInferConfig)


@dataclass
class DecoderConfig:               # schema.py:204-226
    hidden_size: int = 512
    num_hidden_layers: int = 8
    num_attention_heads: int = 8
    intermediate_size: int = 2048
    max_position_embeddings: int = 1024
    num_classes: int = 3
    num_attribute_bins: int = 3
    attribute_emb_dim: int = 64
    pad_class_id: int = 0
    attribute_pad_id: int = 0
    context_num_past_xy_pairs: int = 4
    temperature: float = 0.0
    top_p: float = 0.9
    max_output_tokens: int = 25600
    max_bar_token_limit: int = 512
