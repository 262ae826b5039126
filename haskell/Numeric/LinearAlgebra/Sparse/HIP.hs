{-# LANGUAGE ForeignFunctionInterface, FlexibleContexts, ScopedTypeVariables #-}
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
--   NOT compiled in the authoring image (no GHC there); see haskell/README.md.  The same C ABI is exercised through
--   ctypes and the C++ mirror by the test-suite.
module Numeric.LinearAlgebra.Sparse.HIP
  ( linSolve0, LinSolveMethod(..), (#>), (<.>), norm2, (##), (##^), arnoldi, (<\>), triLowerSolve, triUpperSolve
  , BICGSTAB, bicgsInit, bicgstabStep, bicgstabSteps, _xBicgstab, _rBicgstab, _pBicgstab
  , CGS, cgsInit, cgsStep, cgsSteps, _x, _r, _p, _u
  ) where

import Control.Exception (SomeException, evaluate, try)
import Control.Monad.Catch (MonadThrow, throwM)
import Data.IORef
import qualified Data.IntMap.Strict as IM
import Data.Int (Int64)
import Foreign
import Foreign.C.String
import Foreign.C.Types
import System.Environment (lookupEnv)
import System.IO.Unsafe (unsafePerformIO)
import System.Mem.StableName
import System.Mem.Weak (mkWeakPtr)

import Control.Exception.Common (IterationException (..), MatrixException (..), OperandSizeMismatch (..))
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
foreign import ccall safe "&sla_csr_destroy"      p_csr_destroy       :: FunPtr (Ptr Csr -> IO ())
foreign import ccall safe "sla_vec_create"        c_vec_create        :: Ptr Ctx -> Int64 -> Ptr Double -> Ptr (Ptr Vec) -> IO CInt
foreign import ccall safe "&sla_vec_destroy"      p_vec_destroy       :: FunPtr (Ptr Vec -> IO ())
foreign import ccall safe "sla_vec_to_host"       c_vec_to_host       :: Ptr Vec -> Ptr Double -> IO CInt
foreign import ccall safe "sla_spmv"              c_spmv              :: Ptr Csr -> Ptr Vec -> Ptr Vec -> IO CInt
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
(#>) :: R.SpMatrix Double -> R.SpVector Double -> R.SpVector Double
aa #> x = unsafePerformIO $ do
  a <- lower aa; vx <- upload x; vy <- zeros (R.nrows aa)
  withForeignPtr a $ \pa -> withForeignPtr vx $ \px -> withForeignPtr vy $ \py -> c_spmv pa px py >>= check "matVec"
  ys <- downloadList (R.nrows aa) vy
  return (R.fromListSV (R.nrows aa) (pick (I.keys (R.immSM aa)) (zip [0 ..] ys)))
  where
    pick (k : ks) ((i, y) : rest) | k == i = (i, y) : pick ks rest
                                  | otherwise = pick (k : ks) rest
    pick _ _ = []

(<.>) :: R.SpVector Double -> R.SpVector Double -> Double
v <.> w = unsafePerformIO $ do
  a <- upload v; b <- upload w
  withForeignPtr a $ \pa -> withForeignPtr b $ \pb -> alloca $ \out -> c_dot pa pb out >>= check "<.>" >> peek out

norm2 :: R.SpVector Double -> Double
norm2 v = unsafePerformIO $ upload v >>= \a -> withForeignPtr a $ \pa -> alloca $ \out -> c_nrm2 pa out >>= check "norm2" >> peek out

-- | (##) / (##^) (matMat_ AB / ABt, SpMatrix.hs:768-811): structurally dense over present rows x present columns; a size
--   mismatch is the reference's @error "matMat : incompatible matrix sizes"@
(##), (##^) :: R.SpMatrix Double -> R.SpMatrix Double -> R.SpMatrix Double
(##) = matMatWith 0
(##^) = matMatWith 1

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

-- | (<\>) (Class.hs:244-249) as the dead instance defined it (Sparse.hs:1080-1084)
(<\>) :: MonadThrow m => R.SpMatrix Double -> R.SpVector Double -> m (R.SpVector Double)
aa <\> b = pureThrow $ do
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
