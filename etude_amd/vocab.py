"""Token <-> id vocabulary with the reference's file format and API surface.

Mirror of ``etude.data.vocab`` (etude/data/vocab.py:20-222): ``vocab.json`` is
``{"token_to_id": {...}, "special_tokens": [...]}``; bar delimiters are the ordinary tokens
``Bar_BOS`` / ``Bar_EOS``; ``decode_to_event`` int-casts the value of Note/Pos/TimeSig/Duration/Grace
events.  ``EtudeDecoder.generate`` only needs ``get_bar_bos_id``, ``get_bar_eos_id`` and
``decode_sequence_to_events``, so the reference's own ``Vocab`` object works here unchanged as well.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Union

PAD_TOKEN, BOS_TOKEN, EOS_TOKEN, UNK_TOKEN = "<PAD>", "<BOS>", "<EOS>", "<UNK>"
_INT_TYPES = {"Note", "Pos", "TimeSig", "Duration", "Grace"}


@dataclass
class Event:
    type_: str
    value: Union[str, int]

    def __str__(self) -> str:
        return f"{self.type_}_{self.value}"

    def __repr__(self) -> str:
        return f"Event(type={self.type_}, value={self.value})"


class Vocab:
    def __init__(self, special_tokens: List[str] = None):
        self.special_tokens = list(special_tokens) if special_tokens is not None else [PAD_TOKEN, UNK_TOKEN, BOS_TOKEN, EOS_TOKEN]
        self.token_to_id: Dict[str, int] = {}
        self.id_to_token: List[str] = []
        for t in self.special_tokens:
            self._add_token(t)

    def _add_token(self, token: str) -> int:
        if token not in self.token_to_id:
            self.token_to_id[token] = len(self.id_to_token)
            self.id_to_token.append(token)
        return self.token_to_id[token]

    def build_from_events(self, event_sequences) -> None:
        for seq in event_sequences:
            for ev in seq:
                self._add_token(str(ev))

    def encode(self, token) -> int:
        s = str(token)
        tid = self.token_to_id.get(s, self.token_to_id.get(UNK_TOKEN))
        if tid is None:
            raise ValueError(f"Token '{s}' is not in the vocabulary, and no '{UNK_TOKEN}' is defined")
        return tid

    def decode(self, token_id: int) -> str:
        if 0 <= token_id < len(self.id_to_token):
            return self.id_to_token[token_id]
        raise ValueError(f"Invalid token ID: {token_id}")

    def decode_to_event(self, token_id: int) -> Event:
        s = self.decode(token_id)
        if s in self.special_tokens:
            return Event(type_=s, value="")
        try:
            type_, value_str = s.split("_", 1)
            value = int(value_str) if type_ in _INT_TYPES else value_str
        except (ValueError, IndexError):
            type_, value = s, ""
        return Event(type_=type_, value=value)

    def encode_sequence(self, sequence) -> List[int]:
        return [self.encode(t) for t in sequence]

    def decode_sequence(self, ids) -> List[str]:
        pad = self.get_pad_id()
        return [self.decode(i) for i in ids if i != pad]

    def decode_sequence_to_events(self, ids) -> List[Event]:
        pad = self.get_pad_id()
        return [self.decode_to_event(i) for i in ids if i != pad]

    def save(self, filepath) -> None:
        p = Path(filepath)
        p.parent.mkdir(parents=True, exist_ok=True)
        with open(p, "w", encoding="utf-8") as f:
            json.dump({"token_to_id": self.token_to_id, "special_tokens": self.special_tokens}, f, ensure_ascii=False, indent=2)

    @classmethod
    def load(cls, filepath) -> "Vocab":
        p = Path(filepath)
        if not p.exists():
            raise FileNotFoundError(f"Vocabulary file not found: {p}")
        with open(p, "r", encoding="utf-8") as f:
            data = json.load(f)
        inst = cls(special_tokens=data.get("special_tokens", [PAD_TOKEN]))
        inst.token_to_id = data["token_to_id"]
        inst.id_to_token = [""] * len(inst.token_to_id)
        for tok, tid in inst.token_to_id.items():
            inst.id_to_token[tid] = tok
        return inst

    def __len__(self) -> int:
        return len(self.id_to_token)

    def get_pad_id(self) -> int:
        return self.token_to_id.get(PAD_TOKEN, -1)

    def get_bar_bos_id(self) -> int:
        return self.token_to_id.get("Bar_BOS", -1)

    def get_bar_eos_id(self) -> int:
        return self.token_to_id.get("Bar_EOS", -1)
