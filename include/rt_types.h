/* rt_types.h -- codin-free restatement of the base types the hot path touches.
 *
 * The reference includes "codin/codin.h", "codin/linalg.h" and "codin/image.h"
 * (common.h:3-4, scene.h:3-6), a library that is not vendored with it.  These
 * are the layouts inferred from how the reference uses the types:
 *   Slice(T)    .data / .len                       scene.c:28-31
 *   Vec3        .x/.y/.z, .r/.g/.b, .data[]        raytracer.c:169-173, scene.c:213
 *   Vec4        .xyz / .a, .r/.g/.b                raytracer.c:612, driver.c:400-404
 *   Matrix_4x4  .rows[i][j], translation rows[i][3] raytracer.c:670-672,612
 *   Image       components, pixel_type, width, stride, height, pixels
 *                                                   driver.c:747-754
 * Binary compatibility with a real codin build cannot be checked here
 * ("ABI caveat"); this header IS the ABI of this library.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t   u8;
typedef uint8_t   byte;
typedef int32_t   i32;
typedef uint32_t  u32;
typedef int64_t   i64;
typedef uint64_t  u64;
typedef ptrdiff_t isize;
typedef float     f32;
typedef double    f64;
typedef void     *rawptr;

#define Slice(T) struct { T *data; isize len; }

typedef Slice(byte) Byte_Slice;

typedef union {
  struct { f32 x, y; };
  f32 data[2];
} Vec2;

typedef union {
  struct { f32 x, y, z; };
  struct { f32 r, g, b; };
  f32 data[3];
} Vec3;

typedef Vec3 Color3;

typedef union {
  struct { f32 x, y, z, w; };
  struct { f32 r, g, b, a; };
  struct { Vec3 xyz; f32 _w; };
  f32 data[4];
} Vec4;

typedef Vec4 Color4;

typedef struct {
  f32 rows[4][4];
} Matrix_4x4;

typedef enum {
  PT_u8 = 0,
} Pixel_Type;

/* driver.c:747-754 (field names); pixels indexed components*(x + y*stride)+c,
 * raytracer.c:714-716, driver.c:70-87. */
typedef struct {
  isize      components;
  Pixel_Type pixel_type;
  isize      width;
  isize      stride;
  isize      height;
  Byte_Slice pixels;
} Image;

/* scene_init(Scene*, Triangle_Slice, Allocator) (scene.h:101) takes a codin
 * Allocator by value.  Only "give me `size` bytes aligned to `align`" is used
 * on the path (scene.c:84,422).  proc == NULL selects aligned malloc. */
typedef struct {
  rawptr (*proc)(rawptr user, isize size, isize align);
  rawptr user;
} Allocator;

#ifdef __cplusplus
}
#endif

#endif /* RT_TYPES_H */
