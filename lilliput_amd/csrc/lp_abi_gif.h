// lp_abi_gif.h -- what the GIF encoder needs to see of a decoder handle (the reference's encoder reads d->gif directly).
#pragma once
#include "../../include/lilliput_hip.h"
#include "lp_gif.h"

const LpGifReader& lp_gif_reader(giflib_decoder d);
int lp_gif_bg_alpha(giflib_decoder d); // alpha of the canvas background the decoder settled on with its first frame
