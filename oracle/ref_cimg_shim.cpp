// ref_cimg_shim.cpp -- TEST INFRASTRUCTURE (oracle/_ref/libref_cimg.so): the reference's image pyramid computed
// by the GENUINE vendored CImg (/root/reference/thirdLibs/cimg/CImg.h, compiled in place with -Dcimg_display=0,
// never copied into this repo).  Only the calls of reference src/hpmvs/Image.cpp:55-63 are made here:
//
//     for ii: images_[ii] = images_[ii-1].get_resize_halfXY();     (Image.cpp:56-57; mask CImg.h:21189-21203)
//     for ii: images_[ii].permute_axes("cxyz");                     (Image.cpp:62-63)
//
// on a planar CImg<unsigned char>(w, h, 1, 3) exactly like the one CImg::load() returns for an RGB file
// (Image.h:76).  It pins oracle/hpmvs_oracle.c:orc_half_resize and the HIP half_resize_kernel.
#define cimg_display 0
#include <CImg.h>

#include <cstring>

using cimg_library::CImg;

extern "C" {

// rgb: interleaved level-0 image, byte 3*(y*w+x)+c.  out[l] (l = 0..n_levels-1) receives level l in the
// reference's final storage order (after permute_axes("cxyz")); out_w / out_h its size.  Returns 0.
int ref_cimg_pyramid(const unsigned char* rgb, int w, int h, int n_levels, unsigned char** out, int* out_w, int* out_h) {
    if (!rgb || w <= 0 || h <= 0 || n_levels < 1) return -1;
    CImg<unsigned char> level0(w, h, 1, 3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) level0(x, y, 0, c) = rgb[3 * ((size_t)y * w + x) + c];
    cimg_library::CImgList<unsigned char> images(n_levels);
    images[0] = level0;
    for (int ii = 1; ii < n_levels; ii++) images[ii] = images[ii - 1].get_resize_halfXY();
    for (int ii = 0; ii < n_levels; ii++) {
        out_w[ii] = images[ii].width();
        out_h[ii] = images[ii].height();
        images[ii].permute_axes("cxyz");
        if (out[ii]) std::memcpy(out[ii], images[ii].data(), images[ii].size());
    }
    return 0;
}

}  // extern "C"
