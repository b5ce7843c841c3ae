// Short-sequence LSTM with the input projection inside the recurrence (k_lstm_short.hip): DPCRN's intra-frame BiLSTM
// (DPCRN/DPCRN.py:51-54, 65-71: hidden 64 per direction over the 4 frequency rows of every (utterance, frame) pair).
#pragma once
#include <hip/hip_runtime.h>

namespace se {

struct LstmShortArgs {
    // x: element (o, channel c, step t, sequence n) at o * x_o + c * x_c + t * x_t + n; I = 128 channels
    const float* x; long x_o, x_c, x_t;
    // weights per LSTM z (direction): W_ih [4H][I], W_hh [4H][H], bias [4H] (b_ih + b_hh); rows gate-interleaved (row 4u + g)
    const float *wih, *whh, *bias; long wih_z, whh_z, bias_z;
    // out: h_t of unit u at z * out_z + o * out_o + u * out_row + t * out_t + n
    float* out; long out_o, out_z, out_t, out_row;
    int T, S, Z, O, reverse;      // T steps, S sequences per o, O outer items; reverse: bit z set -> LSTM z walks backwards
};
bool lstm_short_supported(int H, int I, int T);
void launch_lstm_short(const LstmShortArgs& a, hipStream_t s);

}  // namespace se
