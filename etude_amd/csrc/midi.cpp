// Standard MIDI file writer for the pipeline's last step -- TinyREMITokenizer.note_to_midi, etude/data/tokenizer.py:499-524
// (called from infer.py:207).  The reference builds `pretty_midi.PrettyMIDI()` with one `Instrument(program=0)` and calls
// `.write()`.  pretty_midi (0.2.x, with mido underneath) is a third-party dependency that is NOT in /root/reference and not
// installed in this image, so this file restates its published write() algorithm for exactly that object:
//   * format-1 file, 220 ticks per beat (PrettyMIDI's default resolution), tempo 120 bpm;
//   * track 0: time signature 4/4 at tick 0, set_tempo 500000 at tick 0, end_of_track one tick later;
//   * track 1: program_change(program 0, channel 0) at tick 0; per note a note_on(velocity) at round(start / tick_scale)
//     and a note_on(velocity 0) at round(end / tick_scale), tick_scale = 60 / (120 * 220) s, Python round() (half to even);
//     events ordered by (tick, kind rank, 256 * pitch + velocity) with a stable sort, end_of_track one tick after the last;
//   * mido's serialisation: variable-length deltas and running status for channel messages.
// PARITY UNPINNED for the byte stream (no pretty_midi here to produce a golden file); tests/test_midi_cpu.py parses the
// file back and checks ticks, order and the header against the algorithm above.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/etude_hip.h"
#include "common.h"

namespace {

struct Ev { long long tick; int rank; int key; unsigned char b0, b1, b2; int len; };

void put_varint(std::vector<unsigned char>& o, unsigned long long v) {
  unsigned char tmp[10]; int n = 0;
  tmp[n++] = (unsigned char)(v & 0x7f);
  while ((v >>= 7)) tmp[n++] = (unsigned char)((v & 0x7f) | 0x80);
  while (n) o.push_back(tmp[--n]);
}
void put_be32(std::vector<unsigned char>& o, uint32_t v) { for (int s = 24; s >= 0; s -= 8) o.push_back((unsigned char)(v >> s)); }
void put_be16(std::vector<unsigned char>& o, uint32_t v) { o.push_back((unsigned char)(v >> 8)); o.push_back((unsigned char)v); }

long long time_to_tick(double t) {
  // PrettyMIDI.time_to_tick on a file-less object: its tick table holds the single time 0.0, so any later time goes through
  // the "past the last known tick" branch, int(round((t - 0.0) / tick_scale)); times <= 0 land on tick 0.
  const double tick_scale = 60.0 / (120.0 * 220);
  if (!(t > 0.0)) return 0;
  return (long long)std::nearbyint(t / tick_scale);      // default rounding mode = half to even = Python's round()
}

}  // namespace

extern "C" int etd_midi_write(const etd_note* notes, long long n, const char* path) {
  if (n < 0 || (n > 0 && !notes) || !path) ETD_FAIL(ETD_EINVAL, "midi_write: bad arguments");
  std::vector<Ev> ev;
  ev.reserve((size_t)2 * n + 1);
  ev.push_back(Ev{0, 6, 0, 0xC0, 0, 0, 2});                                         // program_change, program 0, channel 0
  for (long long i = 0; i < n; ++i) {
    const etd_note& nt = notes[i];
    if (nt.pitch < 0 || nt.pitch > 127 || nt.velocity < 0 || nt.velocity > 127)
      ETD_FAIL(ETD_EINVAL, "midi_write: note %lld has pitch %d / velocity %d outside 0..127 (mido rejects the message)", i, nt.pitch, nt.velocity);
    if (std::isnan(nt.onset) || std::isnan(nt.offset)) ETD_FAIL(ETD_EINVAL, "midi_write: note %lld has a NaN time", i);
    ev.push_back(Ev{time_to_tick(nt.onset), 10, nt.pitch * 256 + nt.velocity, 0x90, (unsigned char)nt.pitch, (unsigned char)nt.velocity, 3});
    ev.push_back(Ev{time_to_tick(nt.offset), 10, nt.pitch * 256, 0x90, (unsigned char)nt.pitch, 0, 3});
  }
  // sorted(track, key=cmp_to_key(event_compare)): by tick, then by the event kind's secondary key; stable.  (The pass that
  // follows in pretty_midi -- "note-off before note-on at the same tick and pitch" -- finds nothing to swap after this
  // ordering, since velocity 0 already sorts first.)
  std::stable_sort(ev.begin(), ev.end(), [](const Ev& a, const Ev& b) {
    if (a.tick != b.tick) return a.tick < b.tick;
    return a.rank * 65536 + a.key < b.rank * 65536 + b.key;
  });
  std::vector<unsigned char> trk1;
  long long prev = 0; int running = -1;
  for (const Ev& e : ev) {
    put_varint(trk1, (unsigned long long)(e.tick - prev)); prev = e.tick;
    if (e.b0 != running) trk1.push_back(e.b0);                                       // mido: running status for channel messages
    trk1.push_back(e.b1);
    if (e.len == 3) trk1.push_back(e.b2);
    running = e.b0;
  }
  put_varint(trk1, 1); trk1.push_back(0xFF); trk1.push_back(0x2F); trk1.push_back(0x00);   // end_of_track at last tick + 1

  // pretty_midi.write() sorts the timing track with the same event_compare as track 1: at equal ticks set_tempo (rank 1) comes
  // before time_signature (rank 2)
  static const unsigned char trk0[] = {0x00, 0xFF, 0x51, 0x03, 0x07, 0xA1, 0x20,             // set_tempo 500000 us/beat (120 bpm)
                                       0x00, 0xFF, 0x58, 0x04, 0x04, 0x02, 0x18, 0x08,       // time_signature 4/4, 24 clocks/click, 8 32nds/beat
                                       0x01, 0xFF, 0x2F, 0x00};                              // end_of_track at tick 1
  std::vector<unsigned char> out;
  out.insert(out.end(), {'M', 'T', 'h', 'd'}); put_be32(out, 6); put_be16(out, 1); put_be16(out, 2); put_be16(out, 220);
  out.insert(out.end(), {'M', 'T', 'r', 'k'}); put_be32(out, (uint32_t)sizeof(trk0)); out.insert(out.end(), trk0, trk0 + sizeof(trk0));
  out.insert(out.end(), {'M', 'T', 'r', 'k'}); put_be32(out, (uint32_t)trk1.size()); out.insert(out.end(), trk1.begin(), trk1.end());
  FILE* f = fopen(path, "wb");
  if (!f) ETD_FAIL(ETD_EIO, "midi_write: cannot open %s", path);
  const size_t w = fwrite(out.data(), 1, out.size(), f);
  if (fclose(f) != 0 || w != out.size()) ETD_FAIL(ETD_EIO, "midi_write: short write to %s", path);
  return ETD_OK;
}
