"""th_rl_amd -- MI355X-native implementation of the iterated-pricing-game hot path of
HakimNessah/th_rl: NoisyPriceState.step + QTable epsilon-greedy sample + replay-buffer
cycle + TD update, fused in hand-written HIP kernels behind the reference's own
Python API (create_game / train_one / QTable / NoisyPriceState / ReplayBuffer).

Layout mirrors the reference package: th_rl_amd.trainer, .agents, .environments,
.buffers, .utils, .main; plus .batched (GameBatch: many games in lockstep) and
csrc/ (the kernels + C ABI, include/thrl.h).
"""
__all__ = ["agents", "environments", "buffers", "trainer", "batched", "utils"]
