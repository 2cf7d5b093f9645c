"""etude_amd -- MI355X-native (gfx950) implementation of Etude's Extract and Decode hot paths.

Drop-in surfaces (same names/signatures as the reference):
    etude_amd.AMTAPC_Extractor      <- etude.data.extractor.AMTAPC_Extractor
    etude_amd.load_etude_decoder    <- etude.utils.model_loader.load_etude_decoder
    etude_amd.EtudeDecoder.generate <- etude.models.etude_decoder.EtudeDecoder.generate
    etude_amd.Vocab / Event         <- etude.data.vocab
    etude_amd.HFT_Transformer       <- etude.models.hft_transformer.HFT_Transformer (prepare.py's transcriber)
    etude_amd.TinyREMITokenizer     <- etude.data.tokenizer.TinyREMITokenizer (native encode / split / decode_to_notes)
All arithmetic runs in libetude_hip.so (hand-written HIP, see csrc/); importing the heavy
modules is lazy so that `import etude_amd` works on a box without a GPU.
"""
import os as _os

# Four decoder engines (run_engines) want four hardware queues that sit on four different compute pipes.  The HIP runtime's
# default of four queues puts the fourth engine's stream on a queue it shares with the null stream: with the default, four engines
# stepping together reach 7.8 engine-steps per ms instead of 10.0 ((history: 4ac2f57) tools/runs/r2_run15.sh vs r2_run16.sh).  The runtime reads
# this when it initialises, so it only takes effect if etude_amd is imported before the first HIP call of the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__all__ = ["AMTAPC_Extractor", "EtudeDecoder", "EtudeDecoderConfig", "load_etude_decoder", "Vocab", "Event",
           "ExtractorConfig", "DecoderConfig", "HFT_Transformer", "HFTConfig", "TinyREMITokenizer", "run_engines"]


def __getattr__(name):
    if name == "AMTAPC_Extractor":
        from .extractor import AMTAPC_Extractor
        return AMTAPC_Extractor
    if name in ("EtudeDecoder", "EtudeDecoderConfig", "load_etude_decoder", "run_engines"):
        from . import decoder
        return getattr(decoder, name)
    if name in ("Vocab", "Event"):
        from . import vocab
        return getattr(vocab, name)
    if name == "TinyREMITokenizer":
        from .tokenizer import TinyREMITokenizer
        return TinyREMITokenizer
    if name == "HFT_Transformer":
        from .hft_transformer import HFT_Transformer
        return HFT_Transformer
    if name in ("ExtractorConfig", "DecoderConfig", "HFTConfig"):
        from . import config
        return getattr(config, name)
    raise AttributeError(name)
