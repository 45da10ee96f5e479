// ref_harness.cpp — TEST INFRASTRUCTURE.  Thin extern "C" shim that compiles the REFERENCE's
// own CPU headers, unmodified and in place under /root/reference/src (include order of
// reference src/main.cu:1-7 minus the CUDA files), into oracle/_ref/libref_*.so.
// It contains no reference code: everything is #include'd from the read-only mount at build
// time by oracle/Makefile.  Only tests/ and the fixture generator load the result.
#include "common.h"
#include "mmio_highlevel.h"
#include "utils.h"
#include "csr2tile.h"
#include "tilespmv_cpu.h"

extern "C" {

int ref_sizeof_value(void) { return (int)sizeof(MAT_VAL_TYPE); }
int ref_sizeof_tile_matrix(void) { return (int)sizeof(Tile_matrix); }

void ref_Tile_create(Tile_matrix *matrix, int rowA, int colA, MAT_PTR_TYPE nnzA,
                     MAT_PTR_TYPE *csrRowPtrA, int *csrColIdxA, MAT_VAL_TYPE *csrValA)
{
    Tile_create(matrix, rowA, colA, nnzA, csrRowPtrA, csrColIdxA, csrValA);
}

void ref_tilespmv_cpu(Tile_matrix *matrix, int *ptroffset1, int *ptroffset2, int *rowblkblock,
                      unsigned int **blkcoostylerowidx, int **blkcoostylerowidx_colstart,
                      int **blkcoostylerowidx_colstop, int rowA, int colA, MAT_PTR_TYPE nnzA,
                      MAT_PTR_TYPE *csrRowPtrA, int *csrColIdxA, MAT_VAL_TYPE *csrValA,
                      MAT_VAL_TYPE *x, MAT_VAL_TYPE *y, MAT_VAL_TYPE *y_golden)
{
    tilespmv_cpu(matrix, ptroffset1, ptroffset2, rowblkblock, blkcoostylerowidx,
                 blkcoostylerowidx_colstart, blkcoostylerowidx_colstop, rowA, colA, nnzA,
                 csrRowPtrA, csrColIdxA, csrValA, x, y, y_golden);
}

int ref_mmio_allinone(int *m, int *n, MAT_PTR_TYPE *nnz, int *isSymmetric,
                      MAT_PTR_TYPE **csrRowPtr, int **csrColIdx, MAT_VAL_TYPE **csrVal,
                      char *filename)
{
    return mmio_allinone(m, n, nnz, isSymmetric, csrRowPtr, csrColIdx, csrVal, filename);
}

void ref_free(void *p) { free(p); }

}  // extern "C"
