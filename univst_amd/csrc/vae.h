// VAE handle internals (see vae.hip).
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/univst.h"
#include "common.h"
#include "unet.h"

struct Vae {
    univst_vae_cfg cfg;
    std::unordered_map<std::string, WTensor> weights, derived;
    std::unordered_map<std::string, float> mix;       // ST-resblock prefix -> time_mixer.mix_factor
    Arena arena;
    bool finalized = false;
    std::string missing;

    ~Vae();
    int load_tensor(const char* key, const void* dev_ptr, int dtype, const int64_t* shape, int ndim, hipStream_t s);
    int finalize(hipStream_t s);
    int reserve(long imgs, int H, int W);
    int decode(const half_t* z, long imgs, int num_frames, int h, int w, half_t* out, hipStream_t s);
    int encode(const half_t* x, long imgs, int H, int W, half_t* moments, hipStream_t s);
    const WTensor* find(const std::string& k) const;
    half_t* W(const std::string& k);
    int missing_error();
};
