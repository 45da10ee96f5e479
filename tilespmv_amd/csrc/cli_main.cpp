// cli_main.cpp — the `test` command line of the reference driver (src/main.cu:15-205), same
// argv and the same stdout lines in the same order, running on the HIP engine:
//     ./test -d <device_id> <matrix.mtx>
// argc < 2 -> usage line, exit 0 (:18-22); argv[1] != "-d" -> silent exit 0 (:47); the file
// name is argv[3] (:58).  Matrix values are replaced by i % 10 and x by i % 10 (:68-69,:93-97)
// and the last rowA % 16 rows are dropped (:71), exactly like the reference, so that its
// PASS / NO PASS check (1 % relative, :186-197) means the same thing.
// Extension that leaves `-d <int>` untouched: `-d 0,1,2,3` shards the matrix by
// tile-rows over the listed devices (call_tilespmv_hip_multi); an optional 4th argument
// `--combine=none|allgather|allreduce` (default allgather) says what happens to y afterwards.
// `--cache[=prefix]` (4th or 5th argument; default prefix = the .mtx path): parse once — the parsed CSR is kept as
// <prefix>.csr_f64|f32 and the created Tile_matrix as <prefix>.tile_f64|f32; a later run whose .mtx is unchanged (size and
// mtime) reads those instead of tokenising the text and re-tiling (the reference re-parses every time,
// src/mmio_highlevel.h:648-682).  The stdout lines of the reference stay, in order; one "  cache: ..." line is added after each of the two steps.
#include <hip/hip_runtime.h>
#include <sys/time.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "../../include/tilespmv.h"

int main(int argc, char **argv)
{
    if (argc < 2) {
        printf("Run the code by './test matrix.mtx'.\n");
        return 0;
    }
    printf("--------------------------------!!!!!!!!------------------------------------\n");
    if (strcmp(argv[1], "-d") != 0) return 0;
    int device_id = argc > 2 ? atoi(argv[2]) : 0;
    int devices[64], ndev = 0;
    if (argc > 2 && strchr(argv[2], ','))
        for (const char *p = argv[2]; *p && ndev < 64;) {
            devices[ndev++] = atoi(p);
            p = strchr(p, ',');
            if (!p) break;
            p++;
        }
    if (ndev > 0) device_id = devices[0];
    int combine = TILESPMV_Y_ALLGATHER;
    bool use_cache = false, device_build = false;   // --device-build: the tiled matrix is made by Tile_create_device and the plan(s) by tilespmv_plan_create_from_csr (preprocessing on the GPU)
    std::string cache_prefix;
    for (int a = 4; a < argc; a++) {
        if (strcmp(argv[a], "--combine=none") == 0) combine = TILESPMV_Y_SHARDED;
        else if (strcmp(argv[a], "--combine=allgather") == 0) combine = TILESPMV_Y_ALLGATHER;
        else if (strcmp(argv[a], "--combine=allreduce") == 0) combine = TILESPMV_Y_ALLREDUCE;
        else if (strcmp(argv[a], "--cache") == 0) use_cache = true;
        else if (strncmp(argv[a], "--cache=", 8) == 0) { use_cache = true; cache_prefix = argv[a] + 8; }
        else if (strcmp(argv[a], "--device-build") == 0) device_build = true;
        else { fprintf(stderr, "unknown option %s (expected --combine=none|allgather|allreduce, --cache[=prefix] or --device-build)\n", argv[a]); return 1; }
    }
    printf("device_id = %i\n", device_id);
    if (argc < 4) { fprintf(stderr, "usage: %s -d <device_id> <matrix.mtx>\n", argv[0]); return 1; }
    char *filename = argv[3];
    printf("MAT: -------------- %s --------------\n", filename);

    int rowA = 0, colA = 0, isSymmetricA = 0;
    MAT_PTR_TYPE nnzA = 0;
    MAT_PTR_TYPE *csrRowPtrA = NULL; int *csrColIdxA = NULL; MAT_VAL_TYPE *csrValA = NULL;
    timeval t1, t2;
    gettimeofday(&t1, NULL);
    const char *suffix = sizeof(MAT_VAL_TYPE) == 8 ? "f64" : "f32";
    if (use_cache && cache_prefix.empty()) cache_prefix = filename;
    const std::string csr_cache = cache_prefix + ".csr_" + suffix, tile_cache = cache_prefix + ".tile_" + suffix;
    int from_cache = 0;
    int rc = use_cache ? mmio_allinone_cached(&rowA, &colA, &nnzA, &isSymmetricA, &csrRowPtrA, &csrColIdxA, &csrValA, filename, csr_cache.c_str(), &from_cache)
                       : mmio_allinone(&rowA, &colA, &nnzA, &isSymmetricA, &csrRowPtrA, &csrColIdxA, &csrValA, filename);
    gettimeofday(&t2, NULL);
    if (rc != 0) { fprintf(stderr, "cannot read %s (mmio_allinone returned %d)\n", filename, rc); return 1; }
    double time_loadmat = (t2.tv_sec - t1.tv_sec) * 1000.0 + (t2.tv_usec - t1.tv_usec) / 1000.0;
    printf("  input matrix A: ( %i, %i ) nnz = %i\n  loadfile time    = %4.5f sec\n", rowA, colA, nnzA, time_loadmat / 1000.0);
    if (use_cache) printf("  cache: CSR %s %s\n", from_cache == 1 ? "read from" : from_cache == 0 ? "parsed from the text and saved to" : "parsed from the text; could not write", csr_cache.c_str());

    for (int i = 0; i < nnzA; i++) csrValA[i] = i % 10;
    rowA = (rowA / TILESPMV_BLOCK_SIZE) * TILESPMV_BLOCK_SIZE;

    if (hipSetDevice(device_id) != hipSuccess) { fprintf(stderr, "hipSetDevice(%d) failed: no such HIP device\n", device_id); return 2; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { fprintf(stderr, "hipGetDeviceProperties failed\n"); return 2; }
    printf("---------------------------------------------------------------------------------------------\n");
    printf("Device [ %i ] %s @ %4.2f MHz\n", device_id, prop.name[0] ? prop.name : prop.gcnArchName, prop.clockRate * 1e-3f);  // marketing name can be empty in containers

    Tile_matrix *matrixA = (Tile_matrix *)malloc(sizeof(Tile_matrix));
    bool tiles_from_cache = false;
    if (use_cache && from_cache == 1) {   // a tile cache is only trusted next to a fresh CSR cache of the same file, and for this very shape
        int r2 = 0, c2 = 0; MAT_PTR_TYPE z2 = 0;
        gettimeofday(&t1, NULL);
        if (tilespmv_matrix_load(matrixA, &r2, &c2, &z2, tile_cache.c_str()) == 0) {
            if (r2 == rowA && c2 == colA && z2 == nnzA) tiles_from_cache = true;
            else Tile_destroy(matrixA);
        }
        gettimeofday(&t2, NULL);
    }
    if (tiles_from_cache) {
        printf("\n  The number of tile = %i\n", matrixA->tilenum);   // the line Tile_create prints (reference src/csr2tile.h:661)
        printf("  cache: Tile_matrix read from %s in %4.5f sec\n", tile_cache.c_str(), ((t2.tv_sec - t1.tv_sec) * 1000.0 + (t2.tv_usec - t1.tv_usec) / 1000.0) / 1000.0);
    } else {
        gettimeofday(&t1, NULL);
        if (device_build) {
            const int drc = Tile_create_device(matrixA, rowA, colA, nnzA, csrRowPtrA, csrColIdxA, csrValA, 0u);
            if (drc != 0) { fprintf(stderr, "Tile_create_device failed (%d)\n", drc); return 2; }
        } else Tile_create(matrixA, rowA, colA, nnzA, csrRowPtrA, csrColIdxA, csrValA);
        gettimeofday(&t2, NULL);
        if (device_build) printf("  device build: Tile_matrix created on the device (and copied back for the CPU check) in %4.5f sec\n", ((t2.tv_sec - t1.tv_sec) * 1000.0 + (t2.tv_usec - t1.tv_usec) / 1000.0) / 1000.0);
        if (use_cache) {
            const int src = tilespmv_matrix_save(matrixA, rowA, colA, nnzA, tile_cache.c_str());
            printf("  cache: Tile_matrix created in %4.5f sec and %s %s\n", ((t2.tv_sec - t1.tv_sec) * 1000.0 + (t2.tv_usec - t1.tv_usec) / 1000.0) / 1000.0,
                   src == 0 ? "saved to" : "could not be written to", tile_cache.c_str());
        }
    }

    MAT_VAL_TYPE *x = (MAT_VAL_TYPE *)malloc(sizeof(MAT_VAL_TYPE) * colA);
    for (int i = 0; i < colA; i++) x[i] = i % 10;
    MAT_VAL_TYPE *y_golden = (MAT_VAL_TYPE *)malloc(sizeof(MAT_VAL_TYPE) * (rowA + 1));
    for (int i = 0; i < rowA; i++) {
        MAT_VAL_TYPE sum = 0;
        for (int j = csrRowPtrA[i]; j < csrRowPtrA[i + 1]; j++) sum += csrValA[j] * x[csrColIdxA[j]];
        y_golden[i] = sum;
    }
    MAT_VAL_TYPE *y = (MAT_VAL_TYPE *)calloc((size_t)rowA + 1, sizeof(MAT_VAL_TYPE));
    int tilenum = matrixA->tilenum;
    int *ptroffset1 = (int *)calloc((size_t)tilenum + 1, sizeof(int));
    int *ptroffset2 = (int *)calloc((size_t)tilenum + 1, sizeof(int));
    int rowblkblock = 0;
    unsigned int *blkcoostylerowidx; int *blkcoostylerowidx_colstart; int *blkcoostylerowidx_colstop;
    tilespmv_cpu(matrixA, ptroffset1, ptroffset2, &rowblkblock, &blkcoostylerowidx, &blkcoostylerowidx_colstart,
                 &blkcoostylerowidx_colstop, rowA, colA, nnzA, csrRowPtrA, csrColIdxA, csrValA, x, y, y_golden);

    MAT_VAL_TYPE alpha = 1.0;
    if (device_build) setenv("TILESPMV_DEVICE_BUILD", "1", 1);   // (the driver's own switch: call_tilespmv_hip(_multi) then build their plans from the CSR arguments on the device)
    memset(y, 0, sizeof(MAT_VAL_TYPE) * rowA);
    if (ndev > 0) {
        const int mrc = call_tilespmv_hip_multi(filename, matrixA, ptroffset1, ptroffset2, rowblkblock, blkcoostylerowidx,
                                blkcoostylerowidx_colstart, blkcoostylerowidx_colstop, rowA, colA, nnzA, csrRowPtrA,
                                csrColIdxA, csrValA, alpha, x, y, y_golden, ndev, devices, combine);
        if (mrc != 0) return 3;   // (message already on stderr)
    } else
        call_tilespmv_hip(filename, matrixA, ptroffset1, ptroffset2, rowblkblock, blkcoostylerowidx, blkcoostylerowidx_colstart,
                          blkcoostylerowidx_colstop, rowA, colA, nnzA, csrRowPtrA, csrColIdxA, csrValA, alpha, x, y, y_golden);

    int error_count = 0;
    for (int i = 0; i < rowA; i++)
        if (std::fabs(y_golden[i] - y[i]) > 0.01 * std::fabs(y[i])) error_count++;
    if (error_count == 0) std::cout << "Check... PASS!" << std::endl;
    else std::cout << "Check... NO PASS! error_count_cuda = " << error_count << std::endl;

    Tile_destroy(matrixA); free(matrixA);
    free(csrValA); free(csrColIdxA); free(csrRowPtrA);
    free(x); free(y); free(y_golden); free(ptroffset1); free(ptroffset2);
    free(blkcoostylerowidx); free(blkcoostylerowidx_colstart); free(blkcoostylerowidx_colstop);
    return error_count == 0 ? 0 : 4;
}
