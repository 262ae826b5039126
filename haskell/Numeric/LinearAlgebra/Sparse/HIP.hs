{-# LANGUAGE ForeignFunctionInterface, FlexibleContexts, ScopedTypeVariables, TypeFamilies #-}
-- | MI355X backend for the SpMV / CGS / BiCGSTAB / Arnoldi hot path, bound through the C ABI of
--   libsla_hip.so (include/sla_hip.h).  Re-exports the reference's names WITH THE REFERENCE'S SIGNATURES so callers only
--   change an import:
--
--   * every function is pure / @MonadThrow m@-only like the original (Sparse.hs:630, :928, :972, :1016): the GPU work
--     hides behind 'unsafePerformIO', library status codes come back as the reference's own exceptions;
--   * the step functions return a NEW state record (@sla_solver_clone@ + one step on the clone), so
--     @iterate (bicgstabStep aa r0hat) s0 !! 20@ (README.md:222-226) never aliases two elements;
--   * an 'R.SpMatrix' is lowered to its device CSR once: 'lower' memoises per heap object ('StableName').
--
--   Two routes onto the device (round 4):
--
--   * the PLAIN functions of the reference -- 'linSolve0', 'arnoldi', 'bicgsInit' / 'bicgstabStep', 'cgsInit' / 'cgsStep',
--     'cgneInit' / 'cgneStep', 'triLowerSolve' ... -- are re-defined here on @SpMatrix Double@ / @SpVector Double@ under the same
--     names: a caller changes an import;
--   * the CLASS methods -- '(#>)', '(<#)', '(<.>)', 'norm2', '(<\>)' (Class.hs:81-87, 126-153, 224-229, 244-249) -- cannot be
--     re-defined for @SpVector Double@ (the reference's own instances would overlap), so this module brings a vector type of its own,
--     'Dev', with instances of the reference's classes: code written against the classes (@V v =>@, 'LinearSystem' ...) dispatches to
--     the GPU when instantiated at 'Dev' ('toDev' / 'fromDev' at the boundary), and keeps running on the CPU at @SpVector Double@.
--     The monomorphic spellings 'matVecHIP', 'vecMatHIP', 'dotHIP', 'norm2HIP', 'linSolveHIP' are the same calls without the wrapper;
--     no name exported here shadows a class method of Numeric.LinearAlgebra.Class.
--
--   NOT compiled in the authoring image (no GHC there); see haskell/README.md.  The same C ABI is exercised through
--   ctypes and the C++ mirror by the test-suite.
module Numeric.LinearAlgebra.Sparse.HIP
  ( -- * plain functions of the reference, same names and signatures
    linSolve0, LinSolveMethod(..), arnoldi, triLowerSolve, triUpperSolve
  , BICGSTAB, bicgsInit, bicgstabStep, bicgstabSteps, _xBicgstab, _rBicgstab, _pBicgstab
  , CGS, cgsInit, cgsStep, cgsSteps, _x, _r, _p, _u
  , CGNE, cgneInit, cgneStep, cgneSteps, _xCgne, _rCgne, _pCgne
    -- * extension: the reference keeps bcgInit / bcgStep commented out (Sparse.hs:889-909)
  , BCG, bcgInit, bcgStep, bcgSteps, _xBcg, _rBcg, _rHatBcg, _pBcg, _pHatBcg
    -- * the class route: a device-dispatching vector type with the reference's instances
  , Dev(..), toDev, fromDev
    -- * monomorphic spellings of the class methods (no wrapper)
  , matVecHIP, vecMatHIP, dotHIP, norm2HIP, linSolveHIP, matMatHIP, matMatTHIP, transposeHIP
    -- * what the device does with a matrix (typed, round 6)
  , FoldKind(..), foldKindHIP
  ) where

import Control.Exception (SomeException, evaluate, try)
import Control.Monad.Catch (MonadThrow, throwM)
import Data.IORef
import qualified Data.IntMap.Strict as IM
import Data.Int (Int32, Int64)
import Foreign
import Foreign.C.String
import Foreign.C.Types
import System.Environment (lookupEnv)
import System.IO.Unsafe (unsafePerformIO)
import System.Mem.StableName
import System.Mem.Weak (mkWeakPtr)

import Control.Monad.Writer.Class (MonadWriter)
import Control.Exception.Common (IterationException (..), MatrixException (..), OperandSizeMismatch (..))
import qualified Numeric.LinearAlgebra.Class as K    -- the reference's classes: the instances for 'Dev' are below
import qualified Data.Sparse.Internal.IntM as I      -- keys of the row map (IntM.hs:52-53)
import qualified Data.Sparse.SpMatrix as R
import qualified Data.Sparse.SpVector as R
import Numeric.LinearAlgebra.Sparse (LinSolveMethod (..))

data Ctx; data Csr; data Vec; data Solver

foreign import ccall safe "sla_ctx_create"        c_ctx_create        :: CInt -> Ptr (Ptr Ctx) -> IO CInt
foreign import ccall safe "sla_ctx_create_multi"  c_ctx_create_multi  :: CInt -> Ptr CInt -> Ptr (Ptr Ctx) -> IO CInt
foreign import ccall safe "sla_ctx_set_option"    c_ctx_set_option    :: Ptr Ctx -> CString -> CString -> IO CInt
foreign import ccall safe "sla_csr_from_coo"      c_csr_from_coo      :: Ptr Ctx -> Int64 -> Int64 -> Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr Double -> CInt -> Ptr (Ptr Csr) -> IO CInt
foreign import ccall safe "sla_csr_dims"          c_csr_dims          :: Ptr Csr -> Ptr Int64 -> Ptr Int64 -> Ptr Int64 -> Ptr Int64 -> IO CInt
foreign import ccall safe "sla_csr_export"        c_csr_export        :: Ptr Csr -> Ptr Int64 -> Ptr Int64 -> Ptr Double -> IO CInt
foreign import ccall safe "sla_csr_matmat"        c_csr_matmat        :: Ptr Csr -> Ptr Csr -> CInt -> Ptr (Ptr Csr) -> IO CInt
foreign import ccall safe "sla_csr_transpose"     c_csr_transpose     :: Ptr Csr -> Ptr (Ptr Csr) -> IO CInt
foreign import ccall safe "&sla_csr_destroy"      p_csr_destroy       :: FunPtr (Ptr Csr -> IO ())
foreign import ccall safe "sla_vec_create"        c_vec_create        :: Ptr Ctx -> Int64 -> Ptr Double -> Ptr (Ptr Vec) -> IO CInt
foreign import ccall safe "&sla_vec_destroy"      p_vec_destroy       :: FunPtr (Ptr Vec -> IO ())
foreign import ccall safe "sla_vec_to_host"       c_vec_to_host       :: Ptr Vec -> Ptr Double -> IO CInt
foreign import ccall safe "sla_spmv"              c_spmv              :: Ptr Csr -> Ptr Vec -> Ptr Vec -> IO CInt
foreign import ccall safe "sla_spmv_t"            c_spmv_t            :: Ptr Csr -> Ptr Vec -> Ptr Vec -> IO CInt
foreign import ccall safe "sla_dot"               c_dot               :: Ptr Vec -> Ptr Vec -> Ptr Double -> IO CInt
foreign import ccall safe "sla_nrm2"              c_nrm2              :: Ptr Vec -> Ptr Double -> IO CInt
foreign import ccall safe "sla_solver_init"       c_solver_init       :: CInt -> Ptr Csr -> Ptr Vec -> Ptr Vec -> Ptr (Ptr Solver) -> IO CInt
foreign import ccall safe "sla_solver_clone"      c_solver_clone      :: Ptr Solver -> Ptr (Ptr Solver) -> IO CInt
foreign import ccall safe "sla_solver_set_shadow" c_solver_set_shadow :: Ptr Solver -> Ptr Vec -> IO CInt
foreign import ccall safe "sla_solver_step"       c_solver_step       :: Ptr Solver -> CInt -> IO CInt
foreign import ccall safe "sla_solver_get"        c_solver_get        :: Ptr Solver -> CInt -> Ptr Vec -> IO CInt
foreign import ccall safe "&sla_solver_destroy"   p_solver_destroy    :: FunPtr (Ptr Solver -> IO ())
foreign import ccall safe "sla_linsolve0"         c_linsolve0         :: CInt -> Ptr Csr -> Ptr Vec -> Ptr Vec -> Ptr () -> Ptr Vec -> Ptr () -> IO CInt
foreign import ccall safe "sla_arnoldi"           c_arnoldi           :: Ptr Csr -> Ptr Vec -> CInt -> Ptr Double -> Ptr Double -> Ptr CInt -> IO CInt
foreign import ccall safe "sla_linsolve"          c_linsolve          :: Ptr Csr -> Ptr Vec -> Ptr Vec -> Ptr () -> IO CInt
foreign import ccall safe "sla_tri_solve"         c_tri_solve         :: Ptr Csr -> CInt -> Ptr Vec -> Ptr Vec -> Ptr Int64 -> IO CInt
foreign import ccall safe "sla_csr_get_props"     c_csr_get_props     :: Ptr Csr -> Ptr Int32 -> IO CInt
foreign import ccall unsafe "sla_last_error"      c_last_error        :: IO CString

-- | One GPU by default; SLA_GPUS=n makes every matrix / vector / solver of this module span the first n devices of the
--   node (sla_ctx_create_multi: the library fans each call out to one rank per device, RCCL between them).
{-# NOINLINE defaultCtx #-}
defaultCtx :: Ptr Ctx
defaultCtx = unsafePerformIO $ do
  n <- maybe 1 read <$> lookupEnv "SLA_GPUS"
  c <- alloca $ \p -> (if n > 1 then c_ctx_create_multi (fromIntegral (n :: Int)) nullPtr p else c_ctx_create 0 p) >>= check "sla_ctx_create" >> peek p
  -- SLA_REFERENCE_BETA=1: bicgstabStep with the reference's literal beta = (rj1 <.> r0hat) / (r <.> r0hat) * alphaj / omegaj from
  -- the stored rj1 (K4 and K5 as separate kernels) instead of rho' through the linearity identity of the fused sweep (INTEGRATION.md)
  lit <- lookupEnv "SLA_REFERENCE_BETA"
  case lit of
    Just "1" -> withCString "bicg_fuse45" $ \k -> withCString "0" $ \v -> c_ctx_set_option c k v >>= check "sla_ctx_set_option"
    _ -> return ()
  -- SLA_RELAXED_FOLD=1 (opt-in): (#>) on irregular matrices (the tile form: more than 2^18 columns, no band structure) with a row's products added in
  -- relaxed order -- within nnz_i eps sum |a_ij x_j| of the reference's ascending left fold, 17 - 20 % faster on BASELINE config 3a, NOT reproducible
  -- bit for bit from run to run (INTEGRATION.md, "(#>) on irregular matrices and the order of a row's sum").  The default is the fold itself.
  ex <- lookupEnv "SLA_RELAXED_FOLD"
  case ex of
    Just "1" -> withCString "tile_relaxed" $ \k -> withCString "1" $ \v -> c_ctx_set_option c k v >>= check "sla_ctx_set_option"
    _ -> return ()
  return c

-- | status code -> the reference's exception / error (Control/Exception/Common.hs:44-76).  INTEGRATION.md section 2 shows
--   this very function.
check :: String -> CInt -> IO ()
check _ 0 = return ()
check who 1 = c_last_error >>= peekCString >>= \s -> throwM (MatVecSizeMismatchException (who ++ " : " ++ s) (0, 0) 0)
check who 2 = throwM (IterE who "Only BICGSTAB_, CGS_, and CGNE_ are implemented" :: IterationException ())
check _ 3 = error "insertSpMatrix : index out of bounds"
check who 9 = c_last_error >>= peekCString >>= \s -> throwM (NeedsPivoting who s :: MatrixException ())
check who _ = c_last_error >>= peekCString >>= \s -> ioError (userError (who ++ ": " ++ s))

-- | Run an IO action that may throw one of the exceptions above inside any 'MonadThrow' (the reference's constraint).
pureThrow :: MonadThrow m => IO a -> m a
pureThrow io = either (throwM :: MonadThrow m => SomeException -> m a) return (unsafePerformIO (try (io >>= evaluate)))
{-# NOINLINE pureThrow #-}

-- | "Lower once": the device CSR of an 'R.SpMatrix', memoised per heap object.  The table is keyed by the hash of the
--   matrix's 'StableName' (buckets hold the names themselves); the 'ForeignPtr' finalizer releases the device copy when
--   the entry is dropped.  Entries do not outlive their matrix: a weak pointer on the 'R.SpMatrix' (which the table itself
--   does not keep alive -- a 'StableName' is not a reference) removes the entry when the matrix is garbage collected, so a
--   program that builds matrices in a loop does not accumulate device copies.  fromListSM
--   semantics are re-applied by the library (sort, last duplicate wins), so toListSM's descending order is irrelevant.
{-# NOINLINE lowered #-}
lowered :: IORef (IM.IntMap [(StableName (R.SpMatrix Double), ForeignPtr Csr)])
lowered = unsafePerformIO (newIORef IM.empty)

lower :: R.SpMatrix Double -> IO (ForeignPtr Csr)
lower aa = do
  sn <- makeStableName $! aa
  tab <- readIORef lowered
  case lookup sn (IM.findWithDefault [] (hashStableName sn) tab) of
    Just a -> return a
    Nothing -> do
      a <- upload1
      atomicModifyIORef' lowered (\t -> (IM.insertWith (++) (hashStableName sn) [(sn, a)] t, ()))
      _ <- mkWeakPtr aa (Just (evict sn))      -- eviction when the matrix dies (the ForeignPtr finalizer then frees the device CSR)
      return a
  where
    (m, n) = R.dim aa
    (is, js, xs) = unzip3 [(fromIntegral i, fromIntegral j, x) | (i, j, x) <- R.toListSM aa]
    upload1 = withArrayLen is $ \nnz pr -> withArray js $ \pc -> withArray xs $ \pv -> alloca $ \out -> do
      c_csr_from_coo defaultCtx (fromIntegral m) (fromIntegral n) (fromIntegral nnz) pr pc pv 0 out >>= check "fromListSM"
      peek out >>= newForeignPtr p_csr_destroy

evict :: StableName (R.SpMatrix Double) -> IO ()
evict sn = atomicModifyIORef' lowered (\t -> (IM.update dropName (hashStableName sn) t, ()))
  where dropName bucket = case filter ((/= sn) . fst) bucket of { [] -> Nothing; b -> Just b }

upload :: R.SpVector Double -> IO (ForeignPtr Vec)
upload v = withArray (R.toDenseListSV v) $ \p -> alloca $ \out -> do
  c_vec_create defaultCtx (fromIntegral (R.dim v)) p out >>= check "sla_vec_create"
  peek out >>= newForeignPtr p_vec_destroy

zeros :: Int -> IO (ForeignPtr Vec)
zeros n = alloca $ \out -> c_vec_create defaultCtx (fromIntegral n) nullPtr out >>= check "sla_vec_create" >> peek out >>= newForeignPtr p_vec_destroy

downloadList :: Int -> ForeignPtr Vec -> IO [Double]
downloadList n fv = withForeignPtr fv $ \v -> allocaArray n $ \p -> c_vec_to_host v p >>= check "sla_vec_to_host" >> peekArray n p

download :: Int -> ForeignPtr Vec -> IO (R.SpVector Double)
download n fv = R.fromListDenseSV n <$> downloadList n fv

-- | a device CSR handle back as an 'R.SpMatrix' (explicit zeros kept: the structure is part of the value)
liftCsr :: ForeignPtr Csr -> IO (R.SpMatrix Double)
liftCsr fc = withForeignPtr fc $ \c -> alloca $ \pm -> alloca $ \pn -> alloca $ \pz -> do
  c_csr_dims c pm pn pz nullPtr >>= check "dim"
  m <- fromIntegral <$> peek pm; n <- fromIntegral <$> peek pn; nz <- fromIntegral <$> peek pz
  allocaArray (m + 1) $ \rp -> allocaArray (max nz 1) $ \ci -> allocaArray (max nz 1) $ \va -> do
    c_csr_export c rp ci va >>= check "toListSM"
    rps <- map fromIntegral <$> peekArray (m + 1) rp
    cis <- map fromIntegral <$> peekArray nz ci
    vas <- peekArray nz va
    let rowsOf = concat [replicate (e - b) i | (i, b, e) <- zip3 [0 ..] rps (tail rps)]
    return (R.fromListSM (m, n) (zip3 rowsOf cis vas))

-- | linSolve0 (Sparse.hs:1016-1072), the reference's signature: @MonadThrow m@ only
linSolve0 :: MonadThrow m => LinSolveMethod -> R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double -> m (R.SpVector Double)
linSolve0 method aa b x0 = pureThrow $ do
  a <- lower aa; vb <- upload b; vx <- upload x0; vo <- zeros (R.ncols aa)
  withForeignPtr a $ \pa -> withForeignPtr vb $ \pb -> withForeignPtr vx $ \px -> withForeignPtr vo $ \po ->
    c_linsolve0 (fromIntegral (fromEnum method)) pa pb px nullPtr po nullPtr >>= check "linSolve0"
  download (R.ncols aa) vo

-- | (#>) (Common.hs:242-250): the result holds a key for every row present in the matrix and no others.  The present
--   rows come out of the row map's keys in ascending order (O(rows)); the dense device result is walked once beside them.
-- | How @(#>)@ on the lowered matrix adds the products of a row (@sla_fold_kind@, include/sla_hip.h): 'FoldExact' = the reference's ascending left
--   fold bit for bit (Common.hs:247-260), reruns bit-identical; 'FoldRegrouped' = a fixed regrouping of long rows (within the rounding bound, reruns
--   bit-identical); 'FoldRelaxed' = order not fixed from run to run (only after the opt-in @SLA_RELAXED_FOLD=1@: no default form is order-relaxed).
data FoldKind = FoldExact | FoldRegrouped | FoldRelaxed deriving (Eq, Show, Enum)

foldKindHIP :: R.SpMatrix Double -> FoldKind
foldKindHIP aa = unsafePerformIO $ do
  a <- lower aa
  -- sla_csr_props: struct_size, fold, x_exchange, nranks :: int32; rows_local, nnz_local :: int64; rowptr_bits, reserved :: int32  (40 bytes)
  allocaBytes 40 $ \p -> do
    pokeElemOff p 0 (40 :: Int32)
    withForeignPtr a $ \pa -> c_csr_get_props pa p >>= check "sla_csr_get_props"
    k <- peekElemOff p 1
    return (toEnum (fromIntegral k))

matVecHIP :: R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double
matVecHIP aa x = unsafePerformIO $ do
  a <- lower aa; vx <- upload x; vy <- zeros (R.nrows aa)
  withForeignPtr a $ \pa -> withForeignPtr vx $ \px -> withForeignPtr vy $ \py -> c_spmv pa px py >>= check "matVec"
  ys <- downloadList (R.nrows aa) vy
  return (R.fromListSV (R.nrows aa) (pick (I.keys (R.immSM aa)) (zip [0 ..] ys)))
  where
    pick (k : ks) ((i, y) : rest) | k == i = (i, y) : pick ks rest
                                  | otherwise = pick (k : ks) rest
    pick _ _ = []

-- | (<#) = vecMatSD (Common.hs:253-256): @v <# aa@ = transpose aa #> v, without materialising the transpose on the host (the
--   library builds the device transpose of a lowered matrix once, lazily); a key for every COLUMN present in the matrix
vecMatHIP :: R.SpVector Double -> R.SpMatrix Double -> R.SpVector Double
vecMatHIP v aa = unsafePerformIO $ do
  a <- lower aa; vx <- upload v; vy <- zeros (R.ncols aa)
  withForeignPtr a $ \pa -> withForeignPtr vx $ \px -> withForeignPtr vy $ \py -> c_spmv_t pa px py >>= check "vecMat"
  ys <- downloadList (R.ncols aa) vy
  let present = IM.keysSet (IM.unions [IM.fromList [(j, ()) | (_, j, _) <- R.toListSM aa]])
  return (R.fromListSV (R.ncols aa) [(j, y) | (j, y) <- zip [0 ..] ys, j `IM.member` IM.fromSet (const ()) present])

dotHIP :: R.SpVector Double -> R.SpVector Double -> Double
dotHIP v w = unsafePerformIO $ do
  a <- upload v; b <- upload w
  withForeignPtr a $ \pa -> withForeignPtr b $ \pb -> alloca $ \out -> c_dot pa pb out >>= check "<.>" >> peek out

norm2HIP :: R.SpVector Double -> Double
norm2HIP v = unsafePerformIO $ upload v >>= \a -> withForeignPtr a $ \pa -> alloca $ \out -> c_nrm2 pa out >>= check "norm2" >> peek out

-- | (##) / (##^) (matMat_ AB / ABt, SpMatrix.hs:768-811): structurally dense over present rows x present columns; a size
--   mismatch is the reference's @error "matMat : incompatible matrix sizes"@
--   (class methods of MatrixRing on @SpMatrix Double@ in the reference: bound here under monomorphic names)
matMatHIP, matMatTHIP :: R.SpMatrix Double -> R.SpMatrix Double -> R.SpMatrix Double
matMatHIP = matMatWith 0
matMatTHIP = matMatWith 1

-- | transposeSM (SpMatrix.hs:717) as a device sort by (column, row)
transposeHIP :: R.SpMatrix Double -> R.SpMatrix Double
transposeHIP m1 = unsafePerformIO $ do
  a <- lower m1
  t <- withForeignPtr a $ \pa -> alloca $ \out -> do
    c_csr_transpose pa out >>= check "transposeSM"
    peek out >>= newForeignPtr p_csr_destroy
  liftCsr t

matMatWith :: CInt -> R.SpMatrix Double -> R.SpMatrix Double -> R.SpMatrix Double
matMatWith tb m1 m2 = unsafePerformIO $ do
  a <- lower m1; b <- lower m2
  c <- withForeignPtr a $ \pa -> withForeignPtr b $ \pb -> alloca $ \out -> do
    rc <- c_csr_matmat pa pb tb out
    if rc == 1 then c_last_error >>= peekCString >>= error else check "matMat" rc
    peek out >>= newForeignPtr p_csr_destroy
  liftCsr c

-- | solver state records: the device keeps x, r, p (, u); field accessors download on demand.  A record is immutable
--   from Haskell's point of view: nothing in this module steps a handle another value still refers to.
newtype BICGSTAB = BICGSTAB (ForeignPtr Solver, Int)
newtype CGS = CGS (ForeignPtr Solver, Int)

initWith :: CInt -> R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double -> IO (ForeignPtr Solver)
initWith meth aa b x0 = do
  a <- lower aa; vb <- upload b; vx <- upload x0
  withForeignPtr a $ \pa -> withForeignPtr vb $ \pb -> withForeignPtr vx $ \px -> alloca $ \out -> do
    c_solver_init meth pa pb px out >>= check "solver init"
    peek out >>= newForeignPtr p_solver_destroy

-- | clone, (optionally) install the caller's shadow residual, take k steps on the clone, return it
steppedCopy :: String -> Maybe (R.SpVector Double) -> Int -> ForeignPtr Solver -> ForeignPtr Solver
steppedCopy who shadow k fs = unsafePerformIO $ withForeignPtr fs $ \s -> alloca $ \out -> do
  c_solver_clone s out >>= check who
  t <- peek out >>= newForeignPtr p_solver_destroy
  withForeignPtr t $ \pt -> do
    case shadow of
      Just r0hat -> upload r0hat >>= \v -> withForeignPtr v $ \pv -> c_solver_set_shadow pt pv >>= check who
      Nothing -> return ()
    c_solver_step pt (fromIntegral k) >>= check who
  return t
{-# NOINLINE steppedCopy #-}

field :: CInt -> (ForeignPtr Solver, Int) -> R.SpVector Double
field k (fs, n) = unsafePerformIO $ do
  v <- zeros n
  withForeignPtr fs $ \s -> withForeignPtr v $ \pv -> c_solver_get s k pv >>= check "solver get"
  download n v

bicgsInit :: R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double -> BICGSTAB
bicgsInit aa b x0 = BICGSTAB (unsafePerformIO (initWith 4 aa b x0), R.ncols aa)

-- | bicgstabStep aa r0hat state (Sparse.hs:972-981): a NEW record.  (`aa` must be the matrix the record was initialised
--   with -- the device state refers to its lowered copy.)
bicgstabStep :: R.SpMatrix Double -> R.SpVector Double -> BICGSTAB -> BICGSTAB
bicgstabStep _aa r0hat (BICGSTAB (fs, n)) = BICGSTAB (steppedCopy "bicgstabStep" (Just r0hat) 1 fs, n)

-- | @iterate (bicgstabStep aa r0hat) s !! k@ with r0hat = b - A x0 (the README's choice), without materialising the
--   intermediate records: one clone, k steps on the device
bicgstabSteps :: Int -> BICGSTAB -> BICGSTAB
bicgstabSteps k (BICGSTAB (fs, n)) = BICGSTAB (steppedCopy "bicgstabStep" Nothing k fs, n)

_xBicgstab, _rBicgstab, _pBicgstab :: BICGSTAB -> R.SpVector Double
_xBicgstab (BICGSTAB s) = field 0 s; _rBicgstab (BICGSTAB s) = field 1 s; _pBicgstab (BICGSTAB s) = field 2 s

cgsInit :: R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double -> CGS
cgsInit aa b x0 = CGS (unsafePerformIO (initWith 3 aa b x0), R.ncols aa)

-- | cgsStep aa rhat state (Sparse.hs:928-939): a NEW record
cgsStep :: R.SpMatrix Double -> R.SpVector Double -> CGS -> CGS
cgsStep _aa rhat (CGS (fs, n)) = CGS (steppedCopy "cgsStep" (Just rhat) 1 fs, n)

cgsSteps :: Int -> CGS -> CGS
cgsSteps k (CGS (fs, n)) = CGS (steppedCopy "cgsStep" Nothing k fs, n)

_x, _r, _p, _u :: CGS -> R.SpVector Double
_x (CGS s) = field 0 s; _r (CGS s) = field 1 s; _p (CGS s) = field 2 s; _u (CGS s) = field 3 s

-- | CGNE (Sparse.hs:855-878): conjugate gradient on the normal equations.  The reference re-transposes the matrix in every step
--   (@transpose aa #> r@, :878); the device state keeps the lowered transpose.
newtype CGNE = CGNE (ForeignPtr Solver, Int)

cgneInit :: R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double -> CGNE
cgneInit aa b x0 = CGNE (unsafePerformIO (initWith 1 aa b x0), R.ncols aa)

-- | cgneStep aa state (Sparse.hs:870-878): a NEW record
cgneStep :: R.SpMatrix Double -> CGNE -> CGNE
cgneStep _aa (CGNE (fs, n)) = CGNE (steppedCopy "cgneStep" Nothing 1 fs, n)

cgneSteps :: Int -> CGNE -> CGNE
cgneSteps k (CGNE (fs, n)) = CGNE (steppedCopy "cgneStep" Nothing k fs, n)

_xCgne, _rCgne, _pCgne :: CGNE -> R.SpVector Double
_xCgne (CGNE s) = field 0 s; _rCgne (CGNE s) = field 1 s; _pCgne (CGNE s) = field 2 s

-- | BCG (Sparse.hs:886-909): an EXTENSION -- the reference declares the record and keeps @bcgInit@ / @bcgStep@ commented out; the device
--   runs exactly those formulas (one @(#>)@, one @(<#)@, two sweeps per step) with @p0 = r0@, @p0hat = r0hat = r0@.  @linSolve0 BCG_@
--   throws @IterE@ here as it does there (:1031).
newtype BCG = BCG (ForeignPtr Solver, Int)

bcgInit :: R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double -> BCG
bcgInit aa b x0 = BCG (unsafePerformIO (initWith 2 aa b x0), R.ncols aa)

-- | bcgStep aa state (the commented code, Sparse.hs:899-909): a NEW record
bcgStep :: R.SpMatrix Double -> BCG -> BCG
bcgStep _aa (BCG (fs, n)) = BCG (steppedCopy "bcgStep" Nothing 1 fs, n)

bcgSteps :: Int -> BCG -> BCG
bcgSteps k (BCG (fs, n)) = BCG (steppedCopy "bcgStep" Nothing k fs, n)

_xBcg, _rBcg, _rHatBcg, _pBcg, _pHatBcg :: BCG -> R.SpVector Double
_xBcg (BCG s) = field 0 s; _rBcg (BCG s) = field 1 s; _pBcg (BCG s) = field 2 s; _rHatBcg (BCG s) = field 4 s; _pHatBcg (BCG s) = field 5 s

-- ---------------------------------------------------------------------------------------------------------------------------
-- The class route.  'Dev' wraps the reference's own sparse vector; its instances of the reference's classes send the heavy methods
-- to the device and leave the O(n) structural algebra ((^+^), (.*): unions of key sets, Class.hs:57-78 / SpVector.hs:107-114) to
-- the reference's host code, so that structural equality and 'Show' of results stay what they are on the CPU path.
-- (A @newtype@ because @instance LinearVectorSpace (SpVector Double)@ exists in the reference, Common.hs:242-245.)
-- ---------------------------------------------------------------------------------------------------------------------------
newtype Dev = Dev { unDev :: R.SpVector Double } deriving (Eq, Show)

toDev :: R.SpVector Double -> Dev
toDev = Dev

fromDev :: Dev -> R.SpVector Double
fromDev = unDev

instance K.AdditiveGroup Dev where
  zeroV = Dev K.zeroV
  Dev a ^+^ Dev b = Dev (a K.^+^ b)
  negateV (Dev a) = Dev (K.negateV a)
  Dev a ^-^ Dev b = Dev (a K.^-^ b)          -- x ^+^ negateV y, like the class default (Class.hs:69)

instance K.VectorSpace Dev where
  type Scalar Dev = Double
  s .* Dev a = Dev (s K..* a)

instance K.InnerSpace Dev where
  Dev a <.> Dev b = dotHIP a b                -- (<.>), Class.hs:81-83

instance K.Normed Dev where
  type Magnitude Dev = Double
  type RealScalar Dev = Double
  norm1 (Dev a) = K.norm1 a
  norm2Sq (Dev a) = let t = norm2HIP a in t * t
  normP p (Dev a) = K.normP p a
  normalize p (Dev a) = Dev (K.normalize p a)
  normalize2 (Dev a) = Dev ((1 / norm2HIP a) K..* a)   -- normalize2 v = (1 / norm2 v) .* v  (SpVector.hs:127-129)
  norm2 (Dev a) = norm2HIP a

instance K.LinearVectorSpace Dev where
  type MatrixType Dev = R.SpMatrix Double
  aa #> Dev x = Dev (matVecHIP aa x)          -- Class.hs:224-229
  Dev x <# aa = Dev (vecMatHIP x aa)

-- | the class's own signature: @(MonadThrow m, MonadWriter w m) => MatrixType v -> v -> m v@ (Class.hs:244-249).  The writer is
--   not written to (the reference's dead instance, Sparse.hs:1080-1088, logs nothing either).
instance K.LinearSystem Dev where
  aa <\> Dev b = Dev <$> linSolveHIP aa b

-- | arnoldi (Sparse.hs:630-667): Q n x (k+1), H (k+1) x k
arnoldi :: MonadThrow m => R.SpMatrix Double -> R.SpVector Double -> Int -> m (R.SpMatrix Double, R.SpMatrix Double)
arnoldi aa b kn = pureThrow $ do
  a <- lower aa; vb <- upload b
  let n = R.ncols aa
  allocaArray (n * (kn + 1)) $ \pq -> allocaArray ((kn + 1) * kn) $ \ph -> alloca $ \pk ->
    withForeignPtr a $ \pa -> withForeignPtr vb $ \pb -> do
      c_arnoldi pa pb (fromIntegral kn) pq ph pk >>= check "arnoldi"
      k <- fromIntegral <$> peek pk
      q <- peekArray (n * (k + 1)) pq
      h <- peekArray ((kn + 1) * kn) ph
      return ( R.fromListDenseSM n q
             , R.fromListSM (k + 1, k) [(i, j, h !! (j * (kn + 1) + i)) | j <- [0 .. k - 1], i <- [0 .. j + 1]] )

-- | (<\>) (Class.hs:244-249) as the dead instance defined it (Sparse.hs:1080-1084): GMRES from x0 = 0.1 * ones
linSolveHIP :: MonadThrow m => R.SpMatrix Double -> R.SpVector Double -> m (R.SpVector Double)
linSolveHIP aa b = pureThrow $ do
  a <- lower aa; vb <- upload b; vo <- zeros (R.ncols aa)
  withForeignPtr a $ \pa -> withForeignPtr vb $ \pb -> withForeignPtr vo $ \po -> c_linsolve pa pb po nullPtr >>= check "<\\>"
  download (R.ncols aa) vo

-- | triLowerSolve / triUpperSolve (Sparse.hs:750-811): status 9 is the reference's NeedsPivoting (mapped in `check`);
--   the device result is already sparsifySV-ed
triLowerSolve, triUpperSolve :: MonadThrow m => R.SpMatrix Double -> R.SpVector Double -> m (R.SpVector Double)
triLowerSolve = triSolve 0 "triLowerSolve"
triUpperSolve = triSolve 1 "triUpperSolve"

triSolve :: MonadThrow m => CInt -> String -> R.SpMatrix Double -> R.SpVector Double -> m (R.SpVector Double)
triSolve upper who tt b = pureThrow $ do
  t <- lower tt; vb <- upload b; vo <- zeros (R.nrows tt)
  withForeignPtr t $ \pt -> withForeignPtr vb $ \pb -> withForeignPtr vo $ \po -> c_tri_solve pt upper pb po nullPtr >>= check who
  R.sparsifySV <$> download (R.nrows tt) vo
