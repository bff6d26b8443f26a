"""Python model of the host-side MMDVM layer (TEST INFRASTRUCTURE), written from the reference sources independently of
qradiolink_amd/host/mmdvm_wire.cpp: BurstTimer (src/bursttimer.cpp:20-280), gr_mmdvm_sink::work (src/gr/gr_mmdvm_sink.cpp:66-176),
gr_mmdvm_source::work (src/gr/gr_mmdvm_source.cpp:65-243), gr_zero_idle_bursts::work (src/gr/gr_zero_idle_bursts.cpp:45-84)."""
import struct

SAMPLES_PER_SLOT, TIME_PER_SAMPLE, SLOT_TIME = 720, 41667, 30000000
BURST_DELAY = 100000000 * 1000000      # the constructor multiplies its default argument (bursttimer.cpp:27)
MARK_SLOT1, MARK_SLOT2, ZERO_SAMPLES = 0x08, 0x04, 720 * 25 // 24


class Timer:
    def __init__(self):
        self.counter = [0] * 7
        self.base = [0] * 7
        self.last_slot = [0] * 7
        self.init = [False] * 7
        self.slots = [[] for _ in range(7)]
        self.burst_delay = BURST_DELAY

    def set_params(self, burst_delay):
        self.burst_delay = burst_delay      # set_params stores its argument unscaled (bursttimer.cpp:171-178)

    def set_timer(self, value, cn):
        self.counter[cn], self.base[cn], self.init[cn] = 0, value, True

    def check_time(self, cn, time_base_received):
        if not self.slots[cn]:
            return 0
        s = self.slots[cn][0]
        if not time_base_received:
            self.counter[cn] += 1
        t = self.base[cn] + self.counter[cn] * TIME_PER_SAMPLE
        if t >= s[1] and s[2] == 0:
            s[2] += 1
            return s[0]
        if t >= s[1]:
            if s[2] >= SAMPLES_PER_SLOT - 1:
                self.slots[cn].pop(0)
                return 0
            s[2] += 1
        return 0

    def allocate_slot(self, slot_no, cn):
        """returns (nsec, timing or None)"""
        elapsed = self.base[0] + self.counter[0] * TIME_PER_SAMPLE
        timing = None
        if elapsed <= self.last_slot[cn]:
            if cn == 0:
                timing = self.last_slot[cn] - elapsed
            self.last_slot[cn] += SLOT_TIME
        elif self.last_slot[cn] == 0:
            self.last_slot[cn] = elapsed
        elif elapsed - self.last_slot[cn] >= SLOT_TIME:
            self.last_slot[cn] = elapsed
        else:
            self.last_slot[cn] += SLOT_TIME
        nsec = self.last_slot[cn] + self.burst_delay
        self.slots[cn].append([slot_no, nsec, 0])
        return nsec, timing


class Sink:
    def __init__(self, timer, nch):
        self.t, self.nch = timer, nch
        self.data = [[] for _ in range(nch)]
        self.ctrl = [[] for _ in range(nch)]
        self.rssi = [[] for _ in range(nch)]
        self.last_rssi = [0] * nch
        self.slot_counter = [0] * nch

    def work(self, samples, rssi, tags):
        """samples[ch] list of int16, rssi[ch] list of float, tags[ch] list of (offset, secs, fracs); returns [(ch, bytes)]"""
        out = []
        for ch in range(self.nch):
            self.rssi[ch].extend(int(abs(v)) for v in rssi[ch])
            tg = sorted(tags[ch])
            for i, x in enumerate(samples[ch]):
                tbr = False
                if self.slot_counter[ch] > 0:
                    self.slot_counter[ch] += 1
                for off, secs, fracs in tg:
                    if off == i:
                        self.t.set_timer(int(round(float(secs * 1000000000) + fracs * 1e9)), ch)
                        tbr = True
                        break
                control = 0
                slot = self.t.check_time(ch, tbr)
                if slot == 1:
                    control, self.slot_counter[ch] = MARK_SLOT1, 1
                if slot == 2:
                    control, self.slot_counter[ch] = MARK_SLOT2, 1
                self.ctrl[ch].append(control)
                self.data[ch].append(x)
                if self.slot_counter[ch] >= SAMPLES_PER_SLOT:
                    r1 = self.rssi[ch][-1] if self.rssi[ch] else 0
                    r2 = 32767
                    if len(self.rssi[ch]) > 1:
                        self.rssi[ch].pop()
                        r2 = self.rssi[ch][-1]
                    self.last_rssi[ch] = min(r1, r2)
                    self.rssi[ch] = []
                    self.slot_counter[ch] = 0
            if len(self.data[ch]) >= SAMPLES_PER_SLOT:
                n = SAMPLES_PER_SLOT
                msg = struct.pack("<II", n, self.last_rssi[ch]) + bytes(self.ctrl[ch][:n]) + struct.pack("<%dh" % n, *self.data[ch][:n])
                out.append((ch, msg))
                del self.data[ch][:n], self.ctrl[ch][:n]
                self.last_rssi[ch] = 0
        return out


class Source:
    def __init__(self, timer, nch, tdma):
        self.t, self.nch, self.tdma = timer, nch, tdma
        self.sn, self.corr = 2, 0
        self.data = [[] for _ in range(nch)]
        self.ctrl = [[] for _ in range(nch)]

    def work(self, messages):
        """messages[ch] = bytes or b''.  Returns (items, out[ch] list, tags [(ch, offset, is_zero, value)], sleep_ns)"""
        if not all(self.t.init[c] for c in range(self.nch)):
            for c in range(self.nch):
                self.data[c], self.ctrl[c] = [], []
            return (0, None, [], 0) if self.tdma else (SAMPLES_PER_SLOT, [[0] * 720 for _ in range(self.nch)], [], 0)
        for c in range(self.nch):
            m = messages[c]
            if len(m) < 1:
                continue
            n, = struct.unpack_from("<I", m, 0)
            if n > 0:
                self.ctrl[c].extend(m[4:4 + n])
                self.data[c].extend(struct.unpack_from("<%dh" % n, m, 4 + n))
        sleep, self.corr = (self.corr, 0) if self.corr > 0 else (0, self.corr)
        out = [[0] * 720 for _ in range(self.nch)]
        tags = []
        for c in range(self.nch):
            if not self.data[c]:
                self.sn = 1 if self.sn == 2 else 2
                tags.append((c, 0, 1, ZERO_SAMPLES))
                nsec, timing = self.t.allocate_slot(self.sn, c)
                if timing is not None:
                    self.corr = timing
                if nsec > 0 and c == 0:
                    tags.append((c, 710, 0, nsec))
        for c in range(self.nch):
            n = min(len(self.data[c]), 720)
            for i in range(n):
                out[c][i] = self.data[c][i]
                for mark, sn in ((MARK_SLOT1, 1), (MARK_SLOT2, 2)):
                    if self.ctrl[c][i] == mark:
                        self.sn = sn
                        nsec, timing = self.t.allocate_slot(sn, c)
                        if timing is not None:
                            self.corr = timing
                        if nsec > 0 and c == 0:
                            tags.append((c, i, 0, nsec))
            del self.data[c][:n], self.ctrl[c][:n]
        return SAMPLES_PER_SLOT, out, tags, sleep


def zero_runs(tags, chan, items_written, num, den):
    runs = []
    for c, off, is_zero, value in tags:
        if not is_zero or c != chan:
            continue
        start = (2 * (items_written + off) * num + den) // (2 * den)
        if runs and start < runs[-1][0] + runs[-1][1]:
            runs[-1][1] = start + value - runs[-1][0]
        else:
            runs.append([start, value])
    return [tuple(r) for r in runs]
